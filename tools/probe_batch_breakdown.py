import sys, time, json
sys.path.insert(0, ".")
import torch
from autoinst_amd import ncuts_api as api, synth
dev = torch.device("cuda", 0)
K=int(sys.argv[1]) if len(sys.argv) > 1 else 12
data=[]
for k in range(K):
    ch = synth.synthetic_chunk(200000, k, tarl=True)
    data.append((torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev)))
torch.cuda.synchronize()
ctx = api.Context(0)
for rep in range(3):
    t0=time.perf_counter()
    graphs = [api.build_affinity(p, f, alpha=1.0, theta=0.5, gamma=0.0, ctx=ctx) for p, f in data]
    t1=time.perf_counter()
    labs, ngs, st = api.ncuts_labels_batch(graphs, None, 0.03)
    t2=time.perf_counter()
    for g in graphs: g.free()
    t3=time.perf_counter()
    print(json.dumps({"affinity_ms": 1e3*(t1-t0), "ncut_ms": 1e3*(t2-t1), "free_ms": 1e3*(t3-t2), **{k: st[k] for k in ("ms_total","ms_eigen","ms_sweep","ms_rebuild","levels","lanczos_steps")}}), flush=True)
