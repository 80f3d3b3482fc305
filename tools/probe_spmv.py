"""SpMV kernel timing at several graph sizes: python tools/probe_spmv.py"""
import sys, json
sys.path.insert(0, ".")
from autoinst_amd import ncuts_api as api, synth
for n in (5000, 20000, 50000, 200000):
    ch = synth.synthetic_chunk(n, 0, tarl=False)
    g = api.build_affinity(ch["points"], None, alpha=1.0, theta=0.0, gamma=0.0)
    ms, by = api.bench_spmv(g, 200)
    print(json.dumps({"n": n, "nnz": g.nnz, "us": ms * 1e3, "GBps": by / ms / 1e6}), flush=True)
    g.free()
