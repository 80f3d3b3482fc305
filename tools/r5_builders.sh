#!/bin/bash
# round 5: bench.py with builder threads (graphs of the next batches built while the batched cuts run) against none
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd "$R"; out=gpurun_out/r5builders.txt; : > $out
for cfg in "0 2" "1 2" "2 2" "1 3" "1 2 h"; do
  set -- $cfg
  extra="--no-host-inputs"; [ "$3" = "h" ] && extra=""
  line=$(timeout -k 10 400 python bench.py --steps 8 --warmup 2 --builders $1 --in-flight $2 --no-cpu-baseline --no-latency-all $extra 2>$R/gpurun_out/r5builders.err | tail -1)
  echo "builders=$1 threads=$2 $3 $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value", d["value"], "ms/step", d["ms_per_step"], "host", d.get("value_host_inputs"), "frac", d.get("roofline",{}).get("frac"), "solo", d.get("roofline",{}).get("frac_solo"), "hbm", d.get("hbm_in_use_gb"))')" | tee -a $out
done
tail -3 $R/gpurun_out/r5builders.err
