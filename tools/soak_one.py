"""One chunk per call, again and again: labels and counters against the first call.  python tools/soak_one.py R [B] [n_points]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth
from bench import CFG, N_POINTS
R = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
npts = int(sys.argv[3]) if len(sys.argv) > 3 else N_POINTS
dev = torch.device("cuda", 0)
ctx = api.Context(0)
keys = ("lanczos_solves", "null_solves", "spmv_rows", "spmv_nnz", "n_groups", "levels", "max_resid", "lanczos_steps")
out = []
for c0 in range(0, 12, B):
    data = []
    for c in range(c0, c0 + B):
        ch = synth.synthetic_chunk(npts, seed=c, tarl=True)
        data.append((torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev)))
    first = None
    for r in range(R):
        graphs = [api.build_affinity(p, f, alpha=CFG["alpha"], theta=CFG["theta"], gamma=CFG["gamma"], ctx=ctx) for p, f in data]
        try:
            labs, ngs, st = api.ncuts_labels_batch(graphs, None, CFG["T"])
        finally:
            for g in graphs: g.free()
        labs = [np.asarray(l).copy() for l in labs]
        cur = {k: st[k] for k in keys}
        if first is None:
            first = (labs, cur)
        else:
            nd = [int((a != b).sum()) for a, b in zip(first[0], labs)]
            diffk = {k: (first[1][k], cur[k]) for k in keys if first[1][k] != cur[k] and k not in ("lanczos_steps", "levels")}
            if any(nd) or diffk:
                out.append({"chunks": [c0, c0 + B], "r": r, "labels_differ": nd, "counters": diffk, "steps": (first[1]["lanczos_steps"], cur["lanczos_steps"])})
print(json.dumps({"R": R, "B": B, "events": len(out), "list": out[:30]}))
