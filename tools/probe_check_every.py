import sys, time, json, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth
dev = torch.device("cuda", 0)
ch = synth.synthetic_chunk(200_000, 0, tarl=True)
P, T = torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev)
ctx = api.Context(0)
for ce in (16, 8, 4, 12, 24):
    best = 1e9
    for rep in range(4):
        g = api.build_affinity(P, T, alpha=1.0, theta=0.5, gamma=0.0, ctx=ctx)
        t0 = time.perf_counter()
        lab, ng, st = api.ncuts_labels(g, 200_000, 0.03, check_every=ce)
        best = min(best, 1e3 * (time.perf_counter() - t0)); g.free()
    print(json.dumps({"check_every": ce, "ncut_ms": best, "steps": st["lanczos_steps"], "groups": ng}), flush=True)
