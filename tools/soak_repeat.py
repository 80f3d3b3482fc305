"""Same batch again and again: are the labels of a chunk the same every time?   python tools/soak_repeat.py K R [n_points]
K host threads, thread w cuts chunks 12w .. 12w+11 in one call, R times; every result is compared with the thread's first."""
import json, os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth
from bench import CFG, N_POINTS

K = int(sys.argv[1]); R = int(sys.argv[2])
npts = int(sys.argv[3]) if len(sys.argv) > 3 else N_POINTS
B = 12
dev = torch.device("cuda", 0)
ctxs = [api.Context(0) for _ in range(K)]
data = []
for c in range(K * B):
    ch = synth.synthetic_chunk(npts, seed=c, tarl=True)
    data.append((torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev)))
bad, lock = [], threading.Lock()

def th(w):
    first = None
    for r in range(R):
        ids = list(range(w * B, (w + 1) * B))
        graphs = [api.build_affinity(data[i][0], data[i][1], alpha=CFG["alpha"], theta=CFG["theta"], gamma=CFG["gamma"], ctx=ctxs[w]) for i in ids]
        try:
            labs, ngs, st = api.ncuts_labels_batch(graphs, None, CFG["T"])
        finally:
            for g in graphs:
                g.free()
        labs = [np.asarray(l).copy() for l in labs]
        if first is None:
            first = (labs, st["lanczos_steps"])
        else:
            for i, a, b in zip(ids, first[0], labs):
                if not np.array_equal(a, b):
                    with lock:
                        bad.append((w, r, i, int((a != b).sum()), int(a.max()), int(b.max()), int(first[1]), int(st["lanczos_steps"])))

ts = [threading.Thread(target=th, args=(w,)) for w in range(K)]
t0 = time.perf_counter()
for t in ts: t.start()
for t in ts: t.join()
print(json.dumps({"K": K, "R": R, "lockstep": os.environ.get("AI_NCUT_LOCKSTEP"), "s": time.perf_counter() - t0, "mismatches": len(bad), "bad": bad[:20]}))
