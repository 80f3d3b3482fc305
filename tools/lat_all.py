"""One chunk alone, for each of 12 seeds, two passes: mean / per-seed latency (ms) -- python tools/lat_all.py"""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth
dev = torch.device("cuda", 0)
data = []
for k in range(12):
    ch = synth.synthetic_chunk(200_000, k, tarl=True)
    data.append((torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev)))
torch.cuda.synchronize()
ctx = api.Context(0)
res = []
for rep in range(3):
    lat = []
    for p, f in data:
        t0 = time.perf_counter()
        g = api.build_affinity(p, f, alpha=1.0, theta=0.5, gamma=0.0, ctx=ctx)
        lab, ng, st = api.ncuts_labels(g, 200_000, 0.03)
        g.free()
        lat.append(1e3 * (time.perf_counter() - t0))
    res.append(lat)
print(json.dumps({"mean_ms": [sum(l) / len(l) for l in res[1:]], "seed0_ms": [l[0] for l in res[1:]]}))
