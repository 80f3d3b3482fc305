"""Device vs NumPy model of the device algorithm on several seeds / sizes (GPU box).

    python tests/tools/parity_seeds.py 15000 1 2 3 4 5 6
Prints one line per seed: groups, identical partition (and order) or the ARI.
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from autoinst_amd import ncuts_api as api, synth
from oracle import ncuts_ref
import gpu_model

n = int(sys.argv[1])
for seed in [int(x) for x in sys.argv[2:]]:
    for mode, theta, T in (("tarl", 0.5, 0.03), ("spatial", 0.0, 0.075)):
        ch = synth.synthetic_chunk(n, seed, tarl=(mode == "tarl"))
        A = ncuts_ref.affinity_sparse(ch["points"], ch["tarl"], alpha=1.0, theta=theta, gamma=0.0)
        t0 = time.perf_counter()
        got = api.normalized_cut(A, n, np.arange(n), T=T)
        t1 = time.perf_counter()
        exp = gpu_model.normalized_cut_model(A, n, np.arange(n), T=T)
        t2 = time.perf_counter()
        same = len(got) == len(exp) and all(np.array_equal(a, b) for a, b in zip(got, exp))
        la, lb = ncuts_ref.groups_to_labels(got, n), ncuts_ref.groups_to_labels(exp, n)
        print(json.dumps({"n": n, "seed": seed, "mode": mode, "groups": len(got), "groups_model": len(exp), "identical_incl_order": bool(same),
                          "ARI": float(ncuts_ref.adjusted_rand_index(la, lb)), "gpu_s": t1 - t0, "model_s": t2 - t1}), flush=True)
