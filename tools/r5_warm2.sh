#!/bin/bash
# round 5: warm start as the default: the GPU suite, then the bench against AI_FLOW_WARM=0
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd "$R"; O=gpurun_out/r5warm2; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$? $(tail -1 $O/suite.log)"; grep -E "^FAILED|^ERROR|Error" $O/suite.log | head -10
for f in 1 0; do
  line=$(AI_FLOW_WARM=$f timeout -k 10 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$line" > $O/bench_$f.json
  echo "warm=$f $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("value", round(d["value"],1), "host", round(d["value_host_inputs"],1), "one chunk", round(d["single_chunk_latency_ms"],1), "mean", round(d["single_chunk_latency_ms_all"]["mean"],1), "steps mean", round(d["lanczos_steps_all"]["mean"]), "frac", round(r["frac"],3), round(r["frac_solo"],3), round(r["frac_aggregate"],3), "groups", d["groups"])')"
done
