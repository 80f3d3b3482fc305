"""One 200k-point TARL+Spatial chunk, a few repeats (for rocprofv3 runs): python tools/probe_one.py [reps] [batch]"""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
data = []
for k in range(B):
    ch = synth.synthetic_chunk(200_000, k, tarl=True)
    data.append((torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev)))
torch.cuda.synchronize()
ctx = api.Context(0)
for rep in range(reps + 1):
    t0 = time.perf_counter()
    graphs = [api.build_affinity(p, f, alpha=1.0, theta=0.5, gamma=0.0, ctx=ctx) for p, f in data]
    if B == 1:
        lab, ng, st = api.ncuts_labels(graphs[0], 200_000, 0.03)
    else:
        labs, ngs, st = api.ncuts_labels_batch(graphs, None, 0.03)
    for g in graphs: g.free()
    print(json.dumps({"ms": 1e3 * (time.perf_counter() - t0), **{k: st[k] for k in ("ms_total", "ms_eigen", "levels", "lanczos_steps")}}), flush=True)
