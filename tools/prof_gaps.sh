#!/bin/bash
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
O=gpurun_out/r3/pg; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/raw -o p -- python3 tools/probe_stream.py 2 "$@" > $O/run.log 2>&1
python3 tools/trace_gaps.py $O/raw
tail -1 $O/run.log
rm -rf $O/raw
