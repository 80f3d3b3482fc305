"""One 12-chunk call alone on the device, every SpMV launch stamped on the device clock: achieved bytes/s of fk_spmv (solo regime)
and the call's wall time.   python tools/solo_spmv.py [reps]   (AUTOINST_HIP_LIB selects a build for A/B runs)"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth
from bench import CFG, N_POINTS, spmv_bytes
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
ctx = api.Context(0)
data = []
for c in range(12):
    ch = synth.synthetic_chunk(N_POINTS, seed=c, tarl=True)
    data.append((torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev)))
out = []
for r in range(reps + 1):
    graphs = [api.build_affinity(p, f, alpha=CFG["alpha"], theta=CFG["theta"], gamma=CFG["gamma"], ctx=ctx) for p, f in data]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    labs, ngs, st = api.ncuts_labels_batch(graphs, None, CFG["T"], time_spmv="clock" if r % 2 == 0 else False)
    dt = time.perf_counter() - t0
    for g in graphs: g.free()
    if r == 0:
        continue
    rec = {"call_ms": 1e3 * dt, "steps": st["lanczos_steps"]}
    if st["ms_spmv"] > 0:
        b = spmv_bytes(int(st["spmv_rows"]), int(st["spmv_nnz"]), int(st["lanczos_steps"]))
        rec.update({"spmv_ms": st["ms_spmv"], "TBps": b / st["ms_spmv"] / 1e9, "frac": b / st["ms_spmv"] / 1e9 / 8.0})
    out.append(rec)
stamped = [o for o in out if "frac" in o]
plain = [o for o in out if "frac" not in o]
print(json.dumps({"lib": os.environ.get("AUTOINST_HIP_LIB", "default"), "frac_solo": sum(o["frac"] for o in stamped) / max(len(stamped), 1),
                  "spmv_ms": [round(o["spmv_ms"], 2) for o in stamped], "call_ms_unstamped": [round(o["call_ms"], 1) for o in plain],
                  "call_ms_stamped": [round(o["call_ms"], 1) for o in stamped]}))
