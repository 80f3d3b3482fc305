"""CPU-only experiment: where do the oracle (SciPy eigsh) and the device algorithm differ?

"hybrid" = the reference recursion with SciPy's shift-invert eigsh on CONNECTED segments (exactly the
reference's call) and the device's explicit null-space rule on DISCONNECTED ones (where SciPy's
answer is an arbitrary null-space vector).  If hybrid == model (the NumPy model of the device
algorithm, Lanczos on connected segments) then every connected solve led to the same cut, and the
only source of oracle-vs-device differences is SciPy's arbitrary choice in the null space.

    python tools/hybrid_parity.py N [tarl|spatial] [seed]
"""
import sys, time, json
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from scipy.sparse.csgraph import connected_components
from oracle import ncuts_ref
from autoinst_amd import synth
import gpu_model

n = int(sys.argv[1]); mode = sys.argv[2] if len(sys.argv) > 2 else "tarl"
theta, T = (0.5, 0.03) if mode == "tarl" else (0.0, 0.075)
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ch = synth.synthetic_chunk(n, seed, tarl=(mode == "tarl"))
A = ncuts_ref.affinity_sparse(ch["points"], ch["tarl"], alpha=1.0, theta=theta, gamma=0.0)
cnt = {"eigsh": 0, "null": 0}

def hybrid(w, labels):
    m = w.shape[0]
    if m > 2 and labels.shape[0] / (n + 1e-8) > 0.01:
        nc, comp = connected_components(w, directed=False)
        if nc > 1:
            d = np.asarray(w.sum(axis=0)).ravel() + 1.0
            ev = gpu_model.null_vector(w, nc, comp); cnt["null"] += 1
        else:
            _, ev, d = ncuts_ref.fiedler(w); ev = gpu_model.fix_sign(ev); cnt["eigsh"] += 1
        mask, mcut, _ = gpu_model.sweep(ev, d, w)
        if mcut < T:
            return hybrid(w[mask][:, mask], labels[mask]) + hybrid(w[~mask][:, ~mask], labels[~mask])
    return [labels]

t0 = time.time(); gh = hybrid(A, np.arange(n)); th = time.time() - t0
t0 = time.time(); gm = gpu_model.normalized_cut_model(A, n, np.arange(n), T=T); tm = time.time() - t0
lh, lm = ncuts_ref.groups_to_labels(gh, n), ncuts_ref.groups_to_labels(gm, n)
same_order = len(gh) == len(gm) and all(np.array_equal(a, b) for a, b in zip(gh, gm))
print(json.dumps({"n": n, "mode": mode, "seed": seed, "hybrid_groups": len(gh), "model_groups": len(gm), "partition_equal": bool(ncuts_ref.partitions_equal(lh, lm)),
                  "same_group_order": bool(same_order), "ARI": ncuts_ref.adjusted_rand_index(lh, lm), "eigsh_calls": cnt["eigsh"], "null_solves": cnt["null"],
                  "hybrid_s": th, "model_s": tm}))
