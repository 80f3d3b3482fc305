#!/bin/bash
# every kernel of the affinity build (cfg2 + cfg4, 11 builds each) under rocprofv3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pa
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o x -- python3 /root/repo/tools/probe_affinity.py 10 > /tmp/pa.log 2>&1
grep '^{' /tmp/pa.log
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/pa/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0][:70]
    print(f"   {n:70s} calls {r['Calls']:>4s} total_us {float(r['TotalDurationNs'])/1e3:9.1f} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
