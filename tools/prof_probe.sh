#!/bin/bash
# rocprofv3 kernel statistics of tools/probe_one.py <reps> <batch> -> stdout table (top kernels, total kernel time)
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
tag=$1; shift
O=gpurun_out/r3/prof_$tag; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 tools/probe_one.py "$@" > $O/run.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/**/*kernel_stats.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("== $tag $@ : total kernel ms %.1f"%(tot/1e6))
for r in rows[:16]: print("  %-58s calls %7s avg %8.2f us  %5.1f%%"%(r["Name"][:58],r["Calls"],float(r["AverageNs"])/1e3,float(r["Percentage"])))
PY
tail -2 $O/run.log
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
