#!/bin/bash
# round 5: larger batched calls with an admission window (threads, batches per step, chunks per batch, window in chunks)
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd "$R"; out=gpurun_out/r5grid.txt; : > $out
for cfg in "2 2 12 0 8" "2 2 24 12 4" "2 2 36 12 3" "2 2 24 8 4" "1 1 48 16 4" "2 2 48 12 2" "3 3 24 8 3"; do
  set -- $cfg
  line=$(timeout -k 10 400 python bench.py --steps $5 --warmup 1 --in-flight $1 --batches $2 --batch $3 --window-chunks $4 --no-cpu-baseline --no-host-inputs --no-latency-all 2>/dev/null | tail -1)
  echo "$cfg $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("frac_solo"), d.get("lanczos_steps"))')" | tee -a $out
done
