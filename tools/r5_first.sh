#!/bin/bash
# round 5, first GPU call: the GPU suite (with the 75 s soak), one 12-chunk call's phases, kernel statistics of it, a short bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd "$R"; O=gpurun_out/r5a; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; rc=$?; echo "suite rc=$rc $(tail -1 $O/gpu_suite.log)"
[ $rc -ne 0 ] && { tail -40 $O/gpu_suite.log; exit 1; }
AI_NCUT_PHASES=1 timeout -k 10 300 python tools/probe_phases.py > $O/phases.log 2>&1; tail -8 $O/phases.log | cut -c1-900
LINES_=30 tools/prof_phases.sh r5a
timeout -k 10 600 python bench.py --steps 10 --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err || { echo bench failed; tail -5 $O/bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5a/bench_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f host %.1f one-chunk %.1f mean %.1f frac ov %.3f solo %.3f agg %.3f; self_checks %s" % (d["value"], d.get("value_host_inputs") or 0,
      d["single_chunk_latency_ms"], d["single_chunk_latency_ms_all"]["mean"], r["frac_overlapped"], r["frac_solo"], r["frac_aggregate"], d["self_checks"]))
PY
