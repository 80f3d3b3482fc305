#!/bin/bash
# bench.py under GPU_MAX_HW_QUEUES = ... (the runtime maps a process's streams onto that many hardware queues; default 4)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/sweep
for q in ${1:-4 6 8}; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 240 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-inputs --no-latency-all \
    > gpurun_out/sweep/q$q.json 2> gpurun_out/sweep/q$q.err || { echo "q=$q failed"; tail -3 gpurun_out/sweep/q$q.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("gpurun_out/sweep/q$q.json").read().strip().splitlines()[-1])
print("GPU_MAX_HW_QUEUES=$q: %.1f chunks/s, %.1f ms/step, one chunk alone %.1f ms, spmv overlapped frac %.3f" % (d["value"], d["ms_per_step"], d["single_chunk_latency_ms"], d["roofline"]["frac_overlapped"]))
PY
done
