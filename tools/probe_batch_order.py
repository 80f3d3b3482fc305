# experiment: does grouping chunks of similar cost into the same batched call change the throughput?
import sys, time, json, threading, queue, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth, sharding
dev = torch.device("cuda", 0)
N, K, B, M = 200_000, 4, 6, 4
chunks = [synth.synthetic_chunk(N, seed=s, tarl=True) for s in range(M * B)]
data = [(torch.from_numpy(c["points"]).to(dev), torch.from_numpy(c["tarl"]).to(dev)) for c in chunks]
ctxs = [api.Context(0) for _ in range(K)]
# solo cost of every chunk (steps and ms)
cost = []
for i, (p, f) in enumerate(data):
    g = api.build_affinity(p, f, alpha=1.0, theta=0.5, gamma=0.0, ctx=ctxs[0]); api.ncuts_labels(g, N, 0.03); g.free()
    g = api.build_affinity(p, f, alpha=1.0, theta=0.5, gamma=0.0, ctx=ctxs[0])
    t0 = time.perf_counter(); lab, ng, st = api.ncuts_labels(g, N, 0.03); dt = time.perf_counter() - t0; g.free()
    cost.append((dt * 1e3, st["lanczos_steps"], g.nnz))
print("solo ms / steps:", [(round(c[0], 1), c[1]) for c in cost], flush=True)
def run(order, steps=12):
    batches = [order[j:j + B] for j in range(0, len(order), B)]
    work = queue.Queue()
    for s in range(steps):
        for b in batches: work.put(b)
    def worker(w):
        while True:
            try: ids = work.get_nowait()
            except queue.Empty: return
            gs = [api.build_affinity(data[i][0], data[i][1], alpha=1.0, theta=0.5, gamma=0.0, ctx=ctxs[w]) for i in ids]
            api.ncuts_labels_batch(gs, None, 0.03)
            for g in gs: g.free()
    ts = [threading.Thread(target=worker, args=(w,)) for w in range(K)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    torch.cuda.synchronize()
    return steps * len(order) / (time.perf_counter() - t0)
idx = list(range(M * B))
by_cost = sorted(idx, key=lambda i: cost[i][0])
by_steps = sorted(idx, key=lambda i: cost[i][1])
inter = [by_cost[(j % B) * M + j // B] for j in range(M * B)]   # every batch gets one chunk of each cost quartile... (mixed)
for name, order in (("index order", idx), ("sorted by solo ms", by_cost), ("sorted by steps", by_steps), ("interleaved (mixed)", inter), ("index order", idx), ("sorted by solo ms", by_cost)):
    run(order, 2)
    print(name, "%.1f chunks/s" % run(order), flush=True)
