#!/bin/bash
# rocprofv3 kernel statistics of tools/probe_phases.py (three 12-chunk calls alone on the device): tools/prof_phases.sh <tag>
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
tag=$1; O=gpurun_out/r4/pp_$tag; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o p -- python3 tools/probe_phases.py > $O/run.log 2>&1
python3 tools/summarize_prof.py $O/raw $O/k | head -${LINES_:-22}
tail -1 $O/run.log
rm -rf $O/raw
