#!/bin/bash
# threads x batches x batch-size grid of bench.py (short runs, no cpu baseline / host-input legs)
for cfg in "$@"; do
  set -- $cfg
  python bench.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-host-inputs --in-flight $1 --batches $2 --batch $3 > /tmp/bg.log 2>&1
  python3 - "$cfg" <<'PY'
import json, sys
l = [x for x in open("/tmp/bg.log") if x.startswith("{")]
if not l:
    print(sys.argv[1], "FAILED"); print(open("/tmp/bg.log").read()[-500:])
else:
    d = json.loads(l[-1]); print(sys.argv[1], "value %.1f ms_per_step %.1f chunks_per_step %d" % (d["value"], d["ms_per_step"], d["config"]["chunks_per_step"]))
PY
done
