"""CPU-side: how far the DEVICE ALGORITHM (its NumPy model, tests/gpu_model.py -- the HIP path equals it group for
group, tests/test_gpu_fullsize.py) is from the UNMODIFIED oracle on the full-size goldens (tests/golden/full_*.npz,
made by oracle/gen_fullsize.py), and how far the oracle is from ITSELF when ARPACK starts in another state
(the `_w3` / `_w7` fixtures).

    python tests/tools/fullsize_delta.py [pattern]   -> one JSON line per fixture (profiles/r02_fullsize_delta.jsonl)
"""
import glob, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from autoinst_amd import synth
from oracle import metrics_ref, ncuts_ref
from oracle.gen_fullsize import MODES, chunk_for, scoring_gt
import gpu_model

pat = sys.argv[1] if len(sys.argv) > 1 else "full_*"
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", pat + ".npz"))):
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    n, mode, seed = meta["n"], meta["mode"], meta["seed"]
    cfg = MODES[mode]
    ch = chunk_for(n, mode, seed)
    gt = scoring_gt(ch["gt"])
    ref = z["labels"].astype(np.int64)
    out = {"fixture": os.path.basename(path)[:-4], "n": n, "mode": mode, "seed": seed, "prewarm": meta["prewarm"], "perm": meta.get("perm"),
           "oracle_groups": meta["groups"], "oracle_seconds": meta["affinity_seconds"] + meta["normalized_cut_seconds"],
           "oracle_scores": meta["scores"]}
    if meta["prewarm"] == 0 and meta.get("perm") is None:
        A = ncuts_ref.affinity_sparse(ch["points"], ch["tarl"], ch["dino"], alpha=cfg["alpha"], theta=cfg["theta"], gamma=cfg["gamma"])
        t0 = time.perf_counter()
        groups = gpu_model.normalized_cut_model(A, n, np.arange(n), T=cfg["T"])
        lab = ncuts_ref.canonical_labels(ncuts_ref.groups_to_labels(groups, n))
        sc = metrics_ref.score(lab + 1, lab + 1, gt)
        out.update({"who": "device algorithm (NumPy model) vs oracle", "model_seconds": time.perf_counter() - t0, "groups": len(groups),
                    "scores": sc})
    else:
        base = os.path.join(ROOT, "tests", "golden", f"full_{n}_{mode}_{seed}.npz")
        zb = np.load(base)
        lab = ref
        ref = zb["labels"].astype(np.int64)
        mb = json.loads(str(zb["meta"]))
        sc = meta["scores"]
        what = f"after {meta['prewarm']} earlier eigsh calls" if meta["prewarm"] else f"on the same points listed in another order (permutation {meta['perm']})"
        out.update({"who": f"oracle {what} vs the oracle in a fresh interpreter", "groups": meta["groups"],
                    "scores": sc, "oracle_groups": mb["groups"], "oracle_scores": mb["scores"]})
    osc = out["oracle_scores"]
    out["ARI"] = ncuts_ref.adjusted_rand_index(lab, ref)
    out["partition_equal"] = bool(ncuts_ref.partitions_equal(lab, ref))
    out["delta"] = {k: sc[k] - osc[k] for k in ("ap", "S_assoc", "p", "r", "f1")}
    print(json.dumps(out), flush=True)
