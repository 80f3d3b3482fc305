"""Full-size check of the warm start: the 200k-point chunks of the bench (seeds 0 .. n-1, TARL+Spatial) and three tri-modal / spatial
ones, labels with the warm start (default) against AI_FLOW_WARM=0 (hash start).   python tools/warm_vs_hash.py [n]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ctx = api.Context(0)
cases = [(200_000, s, "tarl", 0.03) for s in range(n_seeds)] + [(200_000, 0, "spatial", 0.075), (200_000, 1, "spatial", 0.075), (200_000, 0, "tri", 0.005)]
out = {"chunks": 0, "equal": 0, "steps_warm": 0, "steps_hash": 0, "row_steps_warm": 0, "row_steps_hash": 0, "differ": []}
for n, seed, mode, T in cases:
    ch = synth.synthetic_chunk(n, seed, tarl=mode != "spatial", dino=mode == "tri")
    g = api.build_affinity(ch["points"], ch["tarl"] if mode != "spatial" else None, ch["dino"] if mode == "tri" else None, alpha=1.0,
                           theta=0.0 if mode == "spatial" else 0.5, gamma=0.1 if mode == "tri" else 0.0, ctx=ctx)
    os.environ.pop("AI_FLOW_WARM", None)
    lab1, ng1, st1 = api.ncuts_labels(g, n, T)
    os.environ["AI_FLOW_WARM"] = "0"
    lab0, ng0, st0 = api.ncuts_labels(g, n, T)
    os.environ.pop("AI_FLOW_WARM", None)
    g.free()
    same = bool(ng0 == ng1 and np.array_equal(lab0, lab1))
    out["chunks"] += 1
    out["equal"] += same
    out["steps_warm"] += st1["lanczos_steps"]; out["steps_hash"] += st0["lanczos_steps"]
    out["row_steps_warm"] += st1["spmv_rows"]; out["row_steps_hash"] += st0["spmv_rows"]
    if not same:
        out["differ"].append({"seed": seed, "mode": mode, "groups": [ng1, ng0], "labels_that_differ": int((lab0 != lab1).sum())})
    print(mode, seed, "groups", ng1, ng0, "steps", st1["lanczos_steps"], st0["lanczos_steps"], "equal", same, flush=True)
out["row_steps_ratio"] = out["row_steps_warm"] / out["row_steps_hash"]
out["steps_ratio"] = out["steps_warm"] / out["steps_hash"]
print(json.dumps(out))
