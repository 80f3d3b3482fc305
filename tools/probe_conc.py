"""Throughput with several chunks in flight on one GPU: python tools/probe_conc.py N K"""
import sys, time, json, threading
import numpy as np
sys.path.insert(0, ".")
import torch
from autoinst_amd import ncuts_api as api, synth
n = int(sys.argv[1]); K = int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda", 0)
chunks = []
for k in range(K):
    ch = synth.synthetic_chunk(n, k, tarl=True)
    chunks.append((torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev)))
torch.cuda.synchronize()
ctxs = [api.Context(0) for _ in range(K)]
def work(k, r):
    pts, tarl = chunks[k]
    for _ in range(r):
        g = api.build_affinity(pts, tarl, alpha=1.0, theta=0.5, gamma=0.0, ctx=ctxs[k])
        lab, ng, st = api.ncuts_labels(g, n, 0.03)
        g.free()
def run(r):
    th = [threading.Thread(target=work, args=(k, r)) for k in range(K)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return time.perf_counter() - t0
run(2)
dt = run(reps)
print(json.dumps({"n": n, "concurrency": K, "chunks": K * reps, "seconds": dt, "chunks_per_s": K * reps / dt}), flush=True)
