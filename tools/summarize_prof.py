"""Copy a rocprofv3 --kernel-trace --stats run into profiles/ as a compact summary.

    python tools/summarize_prof.py gpurun_out/prof_xyz profiles/r01_bench
writes <out>_kernel_stats.csv (rocprofv3's own table) and <out>_summary.txt.
"""
import csv, glob, shutil, sys
src, out = sys.argv[1], sys.argv[2]
f = (glob.glob(f"{src}/*/*_kernel_stats.csv") + glob.glob(f"{src}/*_kernel_stats.csv"))[0]
shutil.copy(f, out + "_kernel_stats.csv")
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(out + "_summary.txt", "w") as w:
    w.write(f"rocprofv3 --kernel-trace --stats summary ({f.split('/')[-1]}); total kernel time {tot/1e6:.2f} ms\n")
    w.write(f"{'kernel':58s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'pct':>6s}\n")
    for r in rows[:40]:
        name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0][:58]
        w.write(f"{name:58s} {r['Calls']:>8s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:9.2f} "
                f"{float(r['MinNs'])/1e3:8.2f} {float(r['MaxNs'])/1e3:8.2f} {float(r['Percentage']):6.2f}\n")
print(open(out + "_summary.txt").read())
