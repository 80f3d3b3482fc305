#!/bin/bash
# Several soaks side by side on one GPU (one process each): tools/soak_multi.sh <seconds> <tag=lib[,ENV=VAL...]> ...
# e.g. tools/soak_multi.sh 1000 a=libautoinst_hip.so "b=libautoinst_hip.so,AI_FLOW_GUARD_LOG=/root/repo/gpurun_out/soakm/guard_b.log" t=libautoinst_hip_trace.so  (make -C autoinst_amd/csrc trace)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out/soakm
secs=$1; shift
pids=()
for spec in "$@"; do
  tag=${spec%%=*}; rest=${spec#*=}
  lib=${rest%%,*}; envs=${rest#"$lib"}; envs=${envs#,}
  (
    IFS=, read -ra kv <<< "$envs"
    for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
    export AUTOINST_HIP_LIB=$R/autoinst_amd/$lib AI_SOAK=1 AI_SOAK_SECONDS=$secs AI_SOAK_FAILFILE=$R/gpurun_out/soakm/$tag.failure AI_SOAK_RESULT=$R/gpurun_out/soakm/$tag.result.json
    timeout -k 10 $((secs + 200)) python -m pytest tests/test_gpu_soak.py -m gpu -x -q -s > gpurun_out/soakm/$tag.log 2>&1
  ) &
  pids+=($!)
done
# keep the box from thinking the run is hung
while kill -0 "${pids[@]}" 2>/dev/null; do sleep 60; echo "soaks running: $(date +%T) $(ls gpurun_out/soakm/*.failure 2>/dev/null | wc -l) failure files"; done
for spec in "$@"; do tag=${spec%%=*}; echo "== $tag"; grep -E "soak:|passed|failed" gpurun_out/soakm/$tag.log | cut -c1-200; cat gpurun_out/soakm/$tag.failure 2>/dev/null | cut -c1-600; done
