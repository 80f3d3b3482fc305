#!/bin/bash
# One GPU-box pass that produces every summary under profiles/ for this round (run from the repo root):
#   bench line, rocprofv3 --kernel-trace --stats of the timed regime (2 host threads x 12 chunks) and of one
#   batched call alone (1 x 12), kernel overlap of the timed regime, PMC FETCH_SIZE / WRITE_SIZE passes.
# Large traces are summarised here and deleted; only the small files are merged back.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd "$R"; O=gpurun_out/prof; rm -rf $O; mkdir -p $O
timeout -k 10 900 python bench.py --steps 20 > $O/bench_line.json 2> $O/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ov -o ov -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-inputs --no-latency-all > $O/bench_overlapped_profiled.json 2> $O/ov.log || exit 2
python tools/summarize_prof.py $O/ov $O/bench_200k_overlapped > /dev/null && python tools/trace_busy.py $O/ov 0.3 > $O/bench_200k_overlapped_busy.json
rm -rf $O/ov
rocprofv3 --kernel-trace --stats --output-format csv -d $O/solo -o solo -- python3 bench.py --steps 10 --warmup 2 --in-flight 1 --builders 0 --batch 12 --no-cpu-baseline --no-host-inputs --no-latency-all > $O/bench_solo_profiled.json 2> $O/solo.log || exit 3
python tools/summarize_prof.py $O/solo $O/bench_200k_solo > /dev/null
rm -rf $O/solo
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -o f -- python3 tools/probe_batch_breakdown.py 12 > $O/pmc_f.log 2>&1 || exit 4
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -o w -- python3 tools/probe_batch_breakdown.py 12 > $O/pmc_w.log 2>&1 || exit 5
python tools/pmc_summary.py $O/pf $O/pw $O/pmc_traffic "python tools/probe_batch_breakdown.py 12 (one host thread, 12 chunks per batched call: 200000 12 1, 3 calls)" > /dev/null
rm -rf $O/pf $O/pw
ls -la $O
