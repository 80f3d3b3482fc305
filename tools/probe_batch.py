"""Throughput of batched chunks on one GPU: python tools/probe_batch.py N K [threads]"""
import os, sys, time, json, threading
import numpy as np
sys.path.insert(0, ".")
import torch
from autoinst_amd import ncuts_api as api, synth
n = int(sys.argv[1]); K = int(sys.argv[2]); TH = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda", 0)
data = []
for k in range(K * TH):
    ch = synth.synthetic_chunk(n, k, tarl=True)
    data.append((torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev)))
torch.cuda.synchronize()
ctxs = [api.Context(0) for _ in range(TH)]
def work(t, reps):
    for _ in range(reps):
        graphs = [api.build_affinity(p, f, alpha=1.0, theta=0.5, gamma=0.0, ctx=ctxs[t]) for p, f in data[t * K:(t + 1) * K]]
        labs, ngs, st = api.ncuts_labels_batch(graphs, None, 0.03, check_every=(int(os.environ["CE"]) if "CE" in os.environ else None))
        for g in graphs: g.free()
    return st
def run(reps):
    th = [threading.Thread(target=work, args=(t, reps)) for t in range(TH)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return time.perf_counter() - t0
run(2)
reps = 5
dt = run(reps)
st = work(0, 1)
print(json.dumps({"n": n, "batch": K, "threads": TH, "chunks_per_s": K * TH * reps / dt, "ms_per_batch": 1e3 * dt / reps,
                  "steps": st["lanczos_steps"], "levels": st["levels"], "ms_eigen": st["ms_eigen"]}), flush=True)
