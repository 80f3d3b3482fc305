"""Per-kernel means of a rocprofv3 --pmc pass (csv): python tools/pmc_kernel.py <dir> <kernel substring> [...]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)[0]
want = sys.argv[2:]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    if any(w in n for w in want):
        agg[(n, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (n, c), v in sorted(agg.items()):
    print(f"{n[:40]:40s} {c:26s} launches {len(v):6d} mean {sum(v)/len(v):16.3f} max {max(v):16.3f}")
