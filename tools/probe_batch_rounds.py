import json, os, sys, threading, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from autoinst_amd import ncuts_api as api, synth
from bench import CFG, N_POINTS
B = int(sys.argv[1]); K = 2
dev = torch.device("cuda", 0)
ctxs = [api.Context(0) for _ in range(K)]
data = []
for c in range(K * B):
    ch = synth.synthetic_chunk(N_POINTS, seed=c, tarl=True)
    data.append((torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev)))
def th(w):
    for r in range(6):
        k = (w + r) % K
        t0 = time.perf_counter()
        graphs = [api.build_affinity(p, f, alpha=CFG["alpha"], theta=CFG["theta"], gamma=CFG["gamma"], ctx=ctxs[w]) for p, f in data[k * B:(k + 1) * B]]
        t1 = time.perf_counter()
        labs, ngs, st = api.ncuts_labels_batch(graphs, None, CFG["T"])
        t2 = time.perf_counter()
        for g in graphs: g.free()
        print(json.dumps({"thread": w, "round": r, "build_ms": round(1e3 * (t1 - t0), 1), "cut_ms": round(1e3 * (t2 - t1), 1), "steps": st["lanczos_steps"], "levels": st["levels"], "mem": ctxs[w].mem_info()}), flush=True)
ts = [threading.Thread(target=th, args=(w,)) for w in range(K)]
t0 = time.perf_counter()
for t in ts: t.start()
for t in ts: t.join()
print("chunks/s", K * B * 6 / (time.perf_counter() - t0))
