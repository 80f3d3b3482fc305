"""SpMV (or, with AI_BLOCK_LANCZOS=1, four-vector SpMM) kernel on one whole 200k graph, for PMC passes."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth
ch = synth.synthetic_chunk(200000, 0, tarl=False)
g = api.build_affinity(ch["points"], None, alpha=1.0, theta=0.0, gamma=0.0)
ms, by = api.bench_spmv(g, 50)
print(json.dumps({"n": g.n, "nnz": g.nnz, "us": ms * 1e3, "GBps": by / ms / 1e6}), flush=True)
