#!/bin/bash
# rocprofv3 kernel statistics + device occupancy of tools/probe_one.py <reps> <batch>: tools/prof_one3.sh <tag> <reps> <batch>
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
tag=$1; shift; O=gpurun_out/r3/po_$tag; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o p -- python3 tools/probe_one.py "$@" > $O/run.log 2>&1
python3 tools/summarize_prof.py $O/raw $O/k | head -${LINES_:-16}
python3 tools/trace_busy.py $O/raw 0.5 | tee $O/busy.json
tail -1 $O/run.log
rm -rf $O/raw
