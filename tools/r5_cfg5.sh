#!/bin/bash
# round 5: configs[4] with the float32 early filter against the float64 one; the cfg5 tests first
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd "$R"; O=gpurun_out/r5cfg5; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "cfg5" > $O/tests.log 2>&1; rc=$?; echo "cfg5 tests rc=$rc $(tail -1 $O/tests.log)"
[ $rc -ne 0 ] && { tail -30 $O/tests.log; exit 1; }
for f in 1 0; do
  AI_EIGS_F32_FILTER=$f AI_NCUT_DEBUG=1 timeout -k 10 500 python tests/tools/run_cfg5.py > $O/cfg5_f32_$f.json 2> $O/cfg5_f32_$f.err; echo "f32=$f rc=$?"
  grep "chfsi\] outer\|chfsi\] [0-9]* outer" $O/cfg5_f32_$f.err | cut -c1-220 | head -24
  python -c "
import json,sys
d=json.loads(open('$O/cfg5_f32_$f.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('eigs_seconds','eigs_seconds_second_call','eigs_seconds_third_call','spmm_launches','max_true_residual','orthonormality','second_call_identical')})"
done
