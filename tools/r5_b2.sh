#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd "$R"; O=gpurun_out/r5b2; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "run_chunks or cfg3 or cfg5" > $O/t.log 2>&1; echo "rc=$? $(tail -1 $O/t.log)"; grep -E "Error|error|assert" $O/t.log | head -5
tools/r5_cfg5.sh 2>&1 | grep -v "chfsi\] outer" | tail -6
python tools/run_cfg3.py > $O/cfg3.json 2> $O/cfg3.err; tail -2 $O/cfg3.json | cut -c1-600
