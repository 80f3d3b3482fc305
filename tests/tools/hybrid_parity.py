"""CPU-only experiment: where do the oracle (SciPy eigsh) and the device algorithm differ?

"hybrid" = the reference recursion with SciPy's shift-invert eigsh on CONNECTED segments (exactly the
reference's call) and the split into connected components on DISCONNECTED ones (what the reference's
recursion amounts to there, one component per eigsh call: gpu_model.split_components).  If hybrid == model (the NumPy model of the device
algorithm, Lanczos on connected segments) then every connected solve led to the same cut, and the
only source of oracle-vs-device differences is SciPy's arbitrary choice in the null space.

    python tests/tools/hybrid_parity.py N [tarl|spatial] [seed]
"""
import sys, time, json
import numpy as np
import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
from oracle import ncuts_ref
from autoinst_amd import synth
import gpu_model

n = int(sys.argv[1]); mode = sys.argv[2] if len(sys.argv) > 2 else "tarl"
theta, T = (0.5, 0.03) if mode == "tarl" else (0.0, 0.075)
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ch = synth.synthetic_chunk(n, seed, tarl=(mode == "tarl"))
A = ncuts_ref.affinity_sparse(ch["points"], ch["tarl"], alpha=1.0, theta=theta, gamma=0.0)
cnt = {}
t0 = time.time()
gh = gpu_model.normalized_cut_model(A, n, np.arange(n), T=T, stats=cnt, connected_solver=lambda w: ncuts_ref.fiedler(w)[1])
th = time.time() - t0
t0 = time.time(); gm = gpu_model.normalized_cut_model(A, n, np.arange(n), T=T); tm = time.time() - t0
lh, lm = ncuts_ref.groups_to_labels(gh, n), ncuts_ref.groups_to_labels(gm, n)
same_order = len(gh) == len(gm) and all(np.array_equal(a, b) for a, b in zip(gh, gm))
print(json.dumps({"n": n, "mode": mode, "seed": seed, "hybrid_groups": len(gh), "model_groups": len(gm), "partition_equal": bool(ncuts_ref.partitions_equal(lh, lm)),
                  "same_group_order": bool(same_order), "ARI": ncuts_ref.adjusted_rand_index(lh, lm), "eigsh_calls": cnt.get("eigsh", 0), "component_splits": cnt.get("null", 0),
                  "hybrid_s": th, "model_s": tm}))
