"""Timing probe (GPU box): python tools/probe.py N [tarl|spatial] [seed]"""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from autoinst_amd import ncuts_api as api, synth
n = int(sys.argv[1]); mode = sys.argv[2] if len(sys.argv) > 2 else "tarl"; seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ch = synth.synthetic_chunk(n, seed, tarl=(mode == "tarl"))
theta = 0.5 if mode == "tarl" else 0.0
T = 0.03 if mode == "tarl" else 0.075
api.default_context()
for rep in range(2):
    t0 = time.perf_counter()
    g = api.build_affinity(ch["points"], ch["tarl"], alpha=1.0, theta=theta, gamma=0.0)
    t1 = time.perf_counter()
    lab, ng, st = api.ncuts_labels(g, n, T)
    t2 = time.perf_counter()
    print(json.dumps({"n": n, "mode": mode, "nnz": g.nnz, "affinity_s": t1 - t0, "ncut_s": t2 - t1, "groups": ng, **st}), flush=True)
    if rep == 1:
        ms, by = api.bench_spmv(g, 50)
        print(json.dumps({"spmv_ms": ms, "bytes": by, "GBps": by / ms / 1e6}), flush=True)
    g.free()
