#!/bin/bash
# round 5, after the warm start: is 2 cutting threads x 12 chunks + 1 builder still the best point?  (threads, batches per step, chunks per batch, builders)
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd "$R"; out=gpurun_out/r5grid2.txt; : > $out
for cfg in "2 2 12 1" "2 2 10 1" "2 2 14 1" "2 2 16 1" "3 3 8 1" "3 3 8 2" "2 4 6 1" "2 2 12 1"; do
  set -- $cfg
  line=$(timeout -k 10 400 python bench.py --steps 6 --warmup 2 --in-flight $1 --batches $2 --batch $3 --builders $4 --no-cpu-baseline --no-host-inputs --no-latency-all 2>/dev/null | tail -1)
  echo "$cfg $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("value", round(d["value"],1), "chunks/step", d["config"]["chunks_per_step"], "ms/step", round(d["ms_per_step"],1), "frac", round(r["frac"],3), round(r["frac_solo"],3), round(r["frac_aggregate"],3))')" | tee -a $out
done
