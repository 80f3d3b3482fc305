"""GPU occupancy of a rocprofv3 --kernel-trace run: python tools/trace_busy.py <dir with *_kernel_trace.csv>

Prints the wall span of the trace, the time at least one kernel was running, the mean number of
kernels in flight, and per kernel the time it ran ALONE (nothing else on the device).
"""
import csv, glob, sys, collections, json
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
ev = []
names = {}
for r in csv.DictReader(open(f)):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:40]
    k = names.setdefault(n, len(names))
    ev.append((s, 1, k))
    ev.append((e, -1, k))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0   # ignore the first fraction of the span (set-up)
lo = t0 + skip * (t1 - t0)
busy = 0
conc_int = 0
alone = collections.Counter()
active = collections.Counter()
cur = 0
prev = ev[0][0]
inv = {v: k for k, v in names.items()}
for t, d, k in ev:
    if t > lo and cur > 0:
        dt = t - max(prev, lo)
        busy += dt
        conc_int += dt * cur
        if cur == 1:
            (kk,) = [x for x, c in active.items() if c > 0]
            alone[kk] += dt
    prev = t
    cur += d
    active[k] += d
span = t1 - lo
out = {"span_ms": span / 1e6, "busy_ms": busy / 1e6, "busy_frac": busy / span, "mean_kernels_in_flight_when_busy": conc_int / max(busy, 1),
       "alone_ms": {inv[k]: v / 1e6 for k, v in alone.most_common(8)}}
print(json.dumps(out, indent=1))
