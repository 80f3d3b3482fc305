#!/bin/bash
# round 5: the fused one-launch Lanczos step (AI_FLOW_FUSED=1) against the two-launch step: parity, one 12-chunk call alone, the bench regime
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd "$R"; O=gpurun_out/r5fused; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused" > $O/test.log 2>&1; rc=$?; echo "fused test rc=$rc $(tail -1 $O/test.log)"
[ $rc -ne 0 ] && { tail -40 $O/test.log; exit 1; }
for f in 0 1; do
  AI_FLOW_FUSED=$f AI_NCUT_PHASES=1 timeout -k 10 300 python tools/probe_phases.py > $O/phases_$f.log 2>&1; echo "fused=$f: $(grep '^build' $O/phases_$f.log | tail -1)  $(grep 'chunks 12' $O/phases_$f.log | tail -1 | cut -c1-200)"
done
for f in 0 1; do
  line=$(AI_FLOW_FUSED=$f timeout -k 10 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-inputs 2>/dev/null | tail -1)
  echo "$line" > $O/bench_$f.json
  echo "fused=$f $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("value", d["value"], "ms/step", d["ms_per_step"], "one chunk", d["single_chunk_latency_ms"], "mean", d["single_chunk_latency_ms_all"]["mean"], "frac", r["frac"], "solo", r["frac_solo"], "steps", d.get("lanczos_steps"))')"
done
