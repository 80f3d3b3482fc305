#!/bin/bash
# The round's closing GPU-box call when minutes are short: the GPU suite, the bench line, and the two rocprofv3 summaries (no PMC passes:
# tools/profile_round.sh has those).  Run from the repo root; results under gpurun_out/final.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd "$R"; O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; echo "suite rc=$? $(tail -1 $O/gpu_suite.log)"
timeout -k 10 600 python bench.py --steps 20 > $O/bench_line.json 2> $O/bench.err || { echo bench failed; tail -3 $O/bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final/bench_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f host %.1f ratio %.3f one-chunk %.1f mean %.1f frac ov %.3f solo %.3f agg %.3f copy %.0f" % (d["value"], d.get("value_host_inputs") or 0, d.get("host_inputs_ratio") or 0,
      d["single_chunk_latency_ms"], d["single_chunk_latency_ms_all"]["mean"], r["frac_overlapped"], r["frac_solo"], r["frac_aggregate"], r["device_copy_gbps"]))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ov -o ov -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-inputs --no-latency-all > $O/bench_overlapped_profiled.json 2> $O/ov.log || exit 2
python tools/summarize_prof.py $O/ov $O/bench_200k_overlapped > /dev/null && python tools/trace_busy.py $O/ov 0.3 > $O/bench_200k_overlapped_busy.json
rm -rf $O/ov
rocprofv3 --kernel-trace --stats --output-format csv -d $O/solo -o solo -- python3 bench.py --steps 10 --warmup 2 --in-flight 1 --builders 0 --batch 12 --no-cpu-baseline --no-host-inputs --no-latency-all > $O/bench_solo_profiled.json 2> $O/solo.log || exit 3
python tools/summarize_prof.py $O/solo $O/bench_200k_solo > /dev/null
rm -rf $O/solo
ls $O
