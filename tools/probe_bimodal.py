"""Is the spread of the headline rate (114-123 chunks/s) between processes or inside one?  The 2 x 12 loop timed five times in one
process, 6 rounds each.   python tools/probe_bimodal.py   (run it several times)"""
import json, os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth
from bench import CFG, N_POINTS
dev = torch.device("cuda", 0)
K, B = 2, 12
ctxs = [api.Context(0) for _ in range(K)]
data = []
for c in range(K * B):
    ch = synth.synthetic_chunk(N_POINTS, seed=c, tarl=True)
    data.append((torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev)))
import queue
def loop(rounds):
    q = queue.Queue()
    for r in range(rounds):
        for k in range(K): q.put(k)
    def th(w):
        while True:
            try: k = q.get_nowait()
            except queue.Empty: return
            graphs = [api.build_affinity(p, f, alpha=CFG["alpha"], theta=CFG["theta"], gamma=CFG["gamma"], ctx=ctxs[w]) for p, f in data[k * B:(k + 1) * B]]
            api.ncuts_labels_batch(graphs, None, CFG["T"])
            for g in graphs: g.free()
    ts = [threading.Thread(target=th, args=(w,)) for w in range(K)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    torch.cuda.synchronize()
    return K * B * rounds / (time.perf_counter() - t0)
for w in range(K):
    for k in (0, 1, 0):
        graphs = [api.build_affinity(p, f, alpha=CFG["alpha"], theta=CFG["theta"], gamma=CFG["gamma"], ctx=ctxs[w]) for p, f in data[k * B:(k + 1) * B]]
        api.ncuts_labels_batch(graphs, None, CFG["T"])
        for g in graphs: g.free()
print(json.dumps({"pid": os.getpid(), "chunks_per_s": [round(loop(6), 1) for _ in range(5)]}))
