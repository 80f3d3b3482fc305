#!/bin/bash
# round 5, closing GPU call: the GPU suite, the round's bench line + rocprofv3 summaries + overlap + PMC passes (profile_round.sh),
# configs[0] / [3] (run_cfgs.py) and configs[2] (run_cfg3.py).  Results under gpurun_out/prof and gpurun_out/r5final.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd "$R"; O=gpurun_out/r5final; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; rc=$?; echo "suite rc=$rc $(tail -1 $O/gpu_suite.log)"; tail -1 $O/gpu_suite.log > $O/gpu_suite.txt
[ $rc -ne 0 ] && { tail -40 $O/gpu_suite.log; exit 1; }
cat gpurun_out/soak_result.json 2>/dev/null | tail -1 > $O/suite_soak_result.json
tools/profile_round.sh > $O/profile_round.log 2>&1; echo "profile_round rc=$?"; tail -3 $O/profile_round.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/prof/bench_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f host %.1f ratio %.3f one-chunk %.1f mean %.1f frac ov %.3f solo %.3f agg %.3f copy %.0f idle-blocks %.4f; self_checks %s" % (d["value"], d.get("value_host_inputs") or 0, d.get("host_inputs_ratio") or 0,
      d["single_chunk_latency_ms"], d["single_chunk_latency_ms_all"]["mean"], r["frac_overlapped"], r["frac_solo"], r["frac_aggregate"], r["device_copy_gbps"], r["overlapped"].get("blocks_idle_frac", -1), d["self_checks"]))
PY
python tools/run_cfgs.py > $O/cfgs.jsonl 2> $O/cfgs.err; python tools/run_cfg3.py >> $O/cfgs.jsonl 2>> $O/cfgs.err; python tools/run_cfg3.py --batch 24 >> $O/cfgs.jsonl 2>> $O/cfgs.err; cut -c1-300 $O/cfgs.jsonl
