"""Affinity build time on one MI355X: cfg2 (200k points, 96-d TARL) and cfg4 (+384-d DINO), inputs resident in HBM.

    python tools/probe_affinity.py [reps]      (AI_WEIGHTS_ROWWISE=1 selects the wave-per-row weights kernel)
"""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
ctx = api.Context(0)
ch = synth.synthetic_chunk(200_000, seed=0, tarl=True, dino=True)
P, T, D = (torch.from_numpy(ch[k]).to(dev) for k in ("points", "tarl", "dino"))
torch.cuda.synchronize()
out = {"rowwise": os.environ.get("AI_WEIGHTS_ROWWISE", "0")}
for name, kw in (("cfg2_tarl96", dict(tarl=T, dino=None, theta=0.5, gamma=0.0)), ("cfg4_tarl96_dino384", dict(tarl=T, dino=D, theta=0.5, gamma=0.1))):
    g = api.build_affinity(P, kw["tarl"], kw["dino"], alpha=1.0, theta=kw["theta"], gamma=kw["gamma"], ctx=ctx); nnz = g.nnz; g.free()
    t0 = time.perf_counter()
    for _ in range(reps):
        g = api.build_affinity(P, kw["tarl"], kw["dino"], alpha=1.0, theta=kw["theta"], gamma=kw["gamma"], ctx=ctx); g.free()
    out[name + "_ms"] = 1e3 * (time.perf_counter() - t0) / reps
    out["nnz"] = nnz
print(json.dumps(out))
