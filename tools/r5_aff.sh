#!/bin/bash
# round 5: affinity weights kernel, 32-row tiles (default) against 16-row tiles, parity tests first
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd "$R"; O=gpurun_out/r5aff; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_goldens.py -m gpu -x -q -k "affinity or camera or sam or tarl or symmetric or trimodal or radius" > $O/aff_tests.log 2>&1; rc=$?; echo "affinity tests rc=$rc $(tail -1 $O/aff_tests.log)"
[ $rc -ne 0 ] && { tail -30 $O/aff_tests.log; exit 1; }
tools/prof_affinity.sh "AI_WEIGHTS_TILE=32" "AI_WEIGHTS_TILE=16" 2>&1 | tee $O/prof.txt
