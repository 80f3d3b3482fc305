#!/bin/bash
# rocprofv3 kernel statistics + device occupancy of bench.py in one thread x batch configuration: tools/prof_bench3.sh <tag> <threads> <batches> <batch>
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
tag=$1; O=gpurun_out/r4/pb_$tag; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o p -- python3 bench.py --steps 6 --warmup 2 --in-flight $2 --batches $3 --batch $4 --no-cpu-baseline --no-host-inputs --no-latency-all > $O/line.json 2> $O/err.log
python3 tools/summarize_prof.py $O/raw $O/k | head -24
python3 tools/trace_busy.py $O/raw 0.3 | tee $O/busy.json
python3 -c "import json; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); print('value', d['value'], 'ms_per_step', d['ms_per_step'])"
rm -rf $O/raw
