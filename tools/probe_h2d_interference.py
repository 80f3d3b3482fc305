"""Where does the host-input leg lose its 13-18 %?  The resident loop (2 threads x 12-chunk calls) is timed alone and beside a
background thread that sends pinned host memory to the device in several patterns.   python tools/probe_h2d_interference.py [rounds]"""
import json, os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth
from bench import CFG, N_POINTS

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
only = sys.argv[2] if len(sys.argv) > 2 else None
dev = torch.device("cuda", 0)
K, B = 2, 12
ctxs = [api.Context(0) for _ in range(K)]
data = []
for c in range(K * B):
    ch = synth.synthetic_chunk(N_POINTS, seed=c, tarl=True)
    data.append((torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev)))
hbuf = torch.empty(240_000_000, dtype=torch.float64).pin_memory()      # 1.92 GB = one batch of inputs
dbuf = torch.empty_like(hbuf, device=dev)
cs = torch.cuda.Stream(device=dev)

def batch(w, k):
    mine = data[k * B:(k + 1) * B]
    graphs = [api.build_affinity(p, f, alpha=CFG["alpha"], theta=CFG["theta"], gamma=CFG["gamma"], ctx=ctxs[w]) for p, f in mine]
    try:
        api.ncuts_labels_batch(graphs, None, CFG["T"])
    finally:
        for g in graphs:
            g.free()

def loop(nrounds):
    def th(w):
        for _ in range(nrounds):
            batch(w, w)
    ts = [threading.Thread(target=th, args=(w,)) for w in range(K)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    torch.cuda.synchronize()
    return K * B * nrounds / (time.perf_counter() - t0)

stop = threading.Event()
sent = [0]
def bg(piece_elems, pause_s, rate_GBps):
    """send `piece_elems` doubles at a time; keep the long-run average at rate_GBps (0 = as fast as it goes)"""
    t0 = time.perf_counter()
    with torch.cuda.stream(cs):
        while not stop.is_set():
            off = 0
            while off < hbuf.numel() and not stop.is_set():
                n = min(piece_elems, hbuf.numel() - off)
                dbuf[off:off + n].copy_(hbuf[off:off + n], non_blocking=True)
                off += n
                sent[0] += n * 8
                if pause_s:
                    cs.synchronize(); time.sleep(pause_s)
            cs.synchronize()
            if rate_GBps:
                while sent[0] / (time.perf_counter() - t0) / 1e9 > rate_GBps and not stop.is_set():
                    time.sleep(0.002)

loop(1)
out = {"alone": loop(rounds)}
# the host-input leg's average need is 115 chunks/s * 159 MB = 18 GB/s
for name, args in (("bulk_1.9GB_at_18GBps", (hbuf.numel(), 0, 18.0)), ("bulk_flat_out", (hbuf.numel(), 0, 0)),
                   ("pieces_16MB_at_18GBps", (2_000_000, 0, 18.0)), ("pieces_2MB_paced", (250_000, 0.00005, 18.0)),
                   ("pieces_160MB_at_18GBps", (20_000_000, 0, 18.0))):
    if only and name != only:
        continue
    stop.clear(); sent[0] = 0
    t = threading.Thread(target=bg, args=args); t.start()
    t0 = time.perf_counter()
    v = loop(rounds)
    dt = time.perf_counter() - t0
    stop.set(); t.join()
    out[name] = {"chunks_per_s": v, "copied_GBps": sent[0] / dt / 1e9}
    print(name, out[name], flush=True)
out["alone_again"] = loop(rounds)
print(json.dumps(out))
