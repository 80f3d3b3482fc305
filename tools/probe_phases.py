import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth
from bench import CFG, N_POINTS
dev = torch.device("cuda", 0)
ctx = api.Context(0)
data = []
for c in range(12):
    ch = synth.synthetic_chunk(N_POINTS, seed=c, tarl=True)
    data.append((torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev)))
for r in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    graphs = [api.build_affinity(p, f, alpha=CFG["alpha"], theta=CFG["theta"], gamma=CFG["gamma"], ctx=ctx) for p, f in data]
    torch.cuda.synchronize(); t1 = time.perf_counter()
    labs, ngs, st = api.ncuts_labels_batch(graphs, None, CFG["T"])
    t2 = time.perf_counter()
    for g in graphs: g.free()
    t3 = time.perf_counter()
    print("build %.2f ms, call %.2f ms, free %.2f ms" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2)), flush=True)
