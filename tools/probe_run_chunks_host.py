"""sharding.run_chunks on HOST (NumPy, pageable) inputs against device-resident inputs: chunks/s of 24 chunks of 200k points."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, sharding, synth
from bench import CFG, N_POINTS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda", 0)
host = []
for c in range(n):
    ch = synth.synthetic_chunk(N_POINTS, seed=c % 24, tarl=True)
    host.append((ch["points"], ch["tarl"]))
ctxs = [api.Context(0) for _ in range(2)]
kw = dict(threads=2, batch=12, contexts=ctxs, alpha=CFG["alpha"], theta=CFG["theta"], gamma=CFG["gamma"], T=CFG["T"])
res = {}
ref = None
for name, chunks, extra in (("host_numpy", host, {}),
                            ("device_tensors", [(torch.from_numpy(p).to(dev), torch.from_numpy(f).to(dev)) for p, f in host], {})):
    sharding.run_chunks(chunks, **kw, **extra)
    t0 = time.perf_counter()
    for _ in range(2):
        labs = sharding.run_chunks(chunks, **kw, **extra)
    dt = (time.perf_counter() - t0) / 2
    res[name] = n / dt
    if ref is None:
        ref = labs
    else:
        res["same_labels"] = all(np.array_equal(a, b) for a, b in zip(ref, labs))
print(json.dumps(res))
