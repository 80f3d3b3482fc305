#!/bin/bash
# round 3: threads x batches x batch-size grid of bench.py (24 chunks per step throughout) for the asynchronous frontier
out=${1:-gpurun_out/r3/grid.txt}
for cfg in "4 4 6" "2 2 12" "1 1 24" "2 4 6" "3 3 8" "2 6 4"; do
  set -- $cfg
  line=$(timeout -k 10 300 python bench.py --steps 6 --warmup 2 --in-flight $1 --batches $2 --batch $3 --no-cpu-baseline --no-host-inputs --no-latency-all 2>/dev/null | tail -1)
  echo "$cfg ${AI_NCUT_LOCKSTEP:+lockstep} $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("frac_solo"))')" | tee -a $out
done
