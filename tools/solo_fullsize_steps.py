"""Upper-bound experiments on fk_spmv: 12 roots that never split (T = 0) run exactly `max_iter` full-size steps, every launch stamped.
python tools/solo_fullsize_steps.py [max_iter]   (AUTOINST_HIP_LIB selects the build)"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth, _ffi
from bench import CFG, N_POINTS, spmv_bytes
mi = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
ctx = api.Context(0)
data = []
for c in range(12):
    ch = synth.synthetic_chunk(N_POINTS, seed=c, tarl=True)
    data.append((torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev)))
out = []
for r in range(3):
    graphs = [api.build_affinity(p, f, alpha=CFG["alpha"], theta=CFG["theta"], gamma=CFG["gamma"], ctx=ctx) for p, f in data]
    try:
        labs, ngs, st = api.ncuts_labels_batch(graphs, None, 1e-12, time_spmv="clock", max_iter=mi)
    except _ffi.NoConvergence:
        st = api.last_stats()
    for g in graphs: g.free()
    if r and st["ms_spmv"] > 0:
        b = spmv_bytes(int(st["spmv_rows"]), int(st["spmv_nnz"]), int(st["lanczos_steps"]))
        out.append({"steps": st["lanczos_steps"], "spmv_ms": round(st["ms_spmv"], 2), "us_per_launch": round(1e3 * st["ms_spmv"] / max(st["lanczos_steps"], 1), 2),
                    "frac_by_counters": round(b / st["ms_spmv"] / 1e9 / 8.0, 3)})
print(json.dumps({"lib": os.path.basename(os.environ.get("AUTOINST_HIP_LIB", "default")), "runs": out}))
