#!/bin/bash
# kernel times of the affinity build under rocprofv3, one line per variant:  tools/prof_affinity.sh "<ENV=val ...>" ...
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out
for v in "$@"; do
  rm -rf /tmp/pa
  env $v true
  ( export $v; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o x -- python3 /root/repo/tools/probe_affinity.py 10 > /tmp/pa.log 2>&1 )
  echo "== $v $(grep '^{' /tmp/pa.log)"
  python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/pa/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0]
    if any(k in n for k in ("weights", "neighbours", "zero_rows", "unstash", "count_")):
        print(f"   {n:45s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:9.1f} min {float(r['MinNs'])/1e3:9.1f} max {float(r['MaxNs'])/1e3:9.1f}")
PY
done
