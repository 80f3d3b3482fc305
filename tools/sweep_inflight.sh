#!/bin/bash
# bench.py at other (host threads, chunks per call) points: tools/sweep_inflight.sh "2:12 3:8 3:12 4:6 4:8"
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/sweep
for tb in ${1:-2:12 3:8 3:12 4:6}; do
  t=${tb%%:*}; b=${tb##*:}
  timeout -k 10 240 python bench.py --steps 6 --warmup 2 --in-flight $t --batch $b --no-cpu-baseline --no-host-inputs --no-latency-all \
    > gpurun_out/sweep/t${t}_b${b}.json 2> gpurun_out/sweep/t${t}_b${b}.err || { echo "t=$t b=$b failed"; tail -3 gpurun_out/sweep/t${t}_b${b}.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("gpurun_out/sweep/t${t}_b${b}.json").read().strip().splitlines()[-1])
print("threads $t batch $b: %.1f chunks/s, %.1f ms/step, %d chunks/step, hbm %.1f GB, spmv overlapped frac %.3f" % (d["value"], d["ms_per_step"], d["config"]["chunks_per_step"], d["hbm_in_use_gb"], d["roofline"]["frac_overlapped"]))
PY
done
