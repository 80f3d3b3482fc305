"""Soak of the call workspace (arena) under two host threads: batches of varying composition, labels checked against the
first time each chunk was cut, memory reported as it settles.   python tools/soak_arena.py [batches_per_thread] [n_points]"""
import json, os, random, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth
from bench import CFG, N_POINTS

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
npts = int(sys.argv[2]) if len(sys.argv) > 2 else N_POINTS
dev = torch.device("cuda", 0)
K, NCH = 2, 24
free0, total = torch.cuda.mem_get_info(dev)
ctxs = [api.Context(0) for _ in range(K)]
data = []
for c in range(NCH):
    n = npts if c % 3 else int(npts * (0.55 + 0.02 * c))          # mixed sizes: the workspace need differs call to call
    ch = synth.synthetic_chunk(n, seed=c, tarl=True)
    data.append((torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev), n))
first, lock, bad = {}, threading.Lock(), []

def th(w):
    rng = random.Random(w)
    for r in range(rounds):
        B = rng.choice((4, 8, 12, 12, 12))
        ids = rng.sample(range(NCH), B)
        graphs = [api.build_affinity(data[i][0], data[i][1], alpha=CFG["alpha"], theta=CFG["theta"], gamma=CFG["gamma"], ctx=ctxs[w]) for i in ids]
        try:
            labs, ngs, st = api.ncuts_labels_batch(graphs, [data[i][2] for i in ids], CFG["T"])
        finally:
            for g in graphs:
                g.free()
        with lock:
            for i, l in zip(ids, labs):
                l = np.asarray(l)
                if i not in first:
                    first[i] = l.copy()
                elif not np.array_equal(first[i], l):
                    bad.append((w, r, i, B, int((first[i] != l).sum()), int(first[i].max()), int(l.max())))
        if (r + 1) % 10 == 0:
            f, _ = torch.cuda.mem_get_info(dev)
            print(json.dumps({"thread": w, "batches": r + 1, "ctx": ctxs[w].mem_info(), "hbm_in_use_gb": (total - f) / 1e9}), flush=True)

ts = [threading.Thread(target=th, args=(w,)) for w in range(K)]
t0 = time.perf_counter()
for t in ts: t.start()
for t in ts: t.join()
torch.cuda.synchronize()
f, _ = torch.cuda.mem_get_info(dev)
print(json.dumps({"done": True, "batches": K * rounds, "s": time.perf_counter() - t0, "label_mismatches": len(bad), "bad": bad, "hbm_in_use_gb": (total - f) / 1e9,
                  "ctx": [c.mem_info() for c in ctxs]}))
sys.exit(1 if bad else 0)
