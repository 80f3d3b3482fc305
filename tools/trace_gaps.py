"""Gaps on the stepping stream of a rocprofv3 --kernel-trace run: python tools/trace_gaps.py <dir>
For the kernels fk_spmv / fk_update (the main stream of the asynchronous frontier): time running, gaps between consecutive
ones (start of next - end of previous), which other kernels overlap the gaps."""
import csv, glob, sys, collections, json
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:32]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
main = [(s, e, n) for (s, e, n) in rows if n.startswith("fk_spmv") or n.startswith("fk_update")]
# last call only: take the last 40 % of the main kernels
main = main[int(len(main) * 0.6):]
t0, t1 = main[0][0], main[-1][1]
run = sum(e - s for s, e, _ in main)
gaps = [main[i + 1][0] - main[i][1] for i in range(len(main) - 1)]
big = sorted(gaps)[-20:]
hist = collections.Counter()
for g in gaps:
    hist[min(int(g / 2000), 25)] += 1   # 2 us bins
out = {"span_ms": (t1 - t0) / 1e6, "main_kernels": len(main), "main_running_ms": run / 1e6, "gap_total_ms": sum(gaps) / 1e6,
       "gap_median_us": sorted(gaps)[len(gaps) // 2] / 1e3, "gap_p90_us": sorted(gaps)[int(len(gaps) * 0.9)] / 1e3,
       "largest_gaps_us": [g / 1e3 for g in big], "gap_hist_2us_bins": dict(sorted(hist.items()))}
# what runs during the gaps > 20 us
other = [(s, e, n) for (s, e, n) in rows if not (n.startswith("fk_spmv") or n.startswith("fk_update")) and e > t0 and s < t1]
busy_in_gaps = collections.Counter()
gi = [(main[i][1], main[i + 1][0]) for i in range(len(main) - 1) if main[i + 1][0] - main[i][1] > 20000]
for (gs, ge) in gi[:2000]:
    for (s, e, n) in other:
        if e > gs and s < ge:
            busy_in_gaps[n] += min(e, ge) - max(s, gs)
out["in_gaps_over_20us_ms"] = {k: v / 1e6 for k, v in busy_in_gaps.most_common(8)}
out["gaps_over_20us"] = len(gi)
out["gaps_over_20us_total_ms"] = sum(ge - gs for gs, ge in gi) / 1e6
print(json.dumps(out, indent=1))
