"""CPU experiment: leave one pseudo-randomly composed remainder per disconnected segment un-split, as the reference
does, instead of splitting every component off -- group counts / ARI / score deltas against the full-size oracle goldens."""
import sys, json, glob, numpy as np
import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
from scipy.sparse.csgraph import connected_components
from oracle import ncuts_ref, metrics_ref
from oracle.gen_fullsize import MODES, chunk_for, scoring_gt
import gpu_model as gm

def split_hash(ncomp, comp, labels, n_orig, salt):
    n = comp.shape[0]
    first = np.full(ncomp, n, dtype=np.int64); np.minimum.at(first, comp, np.arange(n))
    sizes = np.bincount(comp, minlength=ncomp)
    key = gm.start_vector(labels[first] * 7919 + salt)       # pseudo-random per component from its first point's original id
    order = np.argsort(key)
    rem, k = n, 0
    while ncomp - k > 1 and gm._eligible(rem, n_orig, 0.01 if k else -1.0):
        rem -= sizes[order[k]]; k += 1
    peeled = [np.flatnonzero(comp == c) for c in order[:k]]
    rest = np.flatnonzero(np.isin(comp, order[k:]))
    return peeled, rest, (ncomp - k == 1)

def model(w, n_orig, labels, T, salt, split_lim=0.01):
    n = w.shape[0]
    if not gm._eligible(n, n_orig, split_lim): return [labels]
    nc, comp = connected_components(w, directed=False)
    if nc > 1:
        peeled, rest, single = split_hash(nc, comp, labels, n_orig, salt)
        out = []
        for idx in peeled: out += model(w[idx][:, idx], n_orig, labels[idx], T, salt)
        out += model(w[rest][:, rest], n_orig, labels[rest], T, salt) if single else [labels[rest]]
        return out
    d = np.asarray(w.sum(axis=0)).ravel() + 1.0
    lam, ev, m, resid = gm.lanczos_fiedler(w, labels)
    mask, mcut, _ = gm.sweep(ev, d, w)
    if mcut < T:
        return model(w[mask][:, mask], n_orig, labels[mask], T, salt) + model(w[~mask][:, ~mask], n_orig, labels[~mask], T, salt)
    return [labels]

for path in sorted(glob.glob(os.path.join(_R, "tests", "golden", "full_*.npz"))):
    z = np.load(path); meta = json.loads(str(z["meta"]))
    if meta["prewarm"] or meta.get("perm") is not None: continue
    if meta["n"] > int(sys.argv[1]): continue
    n, mode, seed = meta["n"], meta["mode"], meta["seed"]; cfg = MODES[mode]
    ch = chunk_for(n, mode, seed); gt = scoring_gt(ch["gt"])
    A = ncuts_ref.affinity_sparse(ch["points"], ch["tarl"], ch["dino"], alpha=1.0, theta=cfg["theta"], gamma=cfg["gamma"])
    ref = z["labels"].astype(np.int64)
    row = [path.split("/")[-1][:-4], "oracle groups", meta["groups"]]
    for salt in (0, 1):
        g = model(A, n, np.arange(n), cfg["T"], salt)
        lab = ncuts_ref.canonical_labels(ncuts_ref.groups_to_labels(g, n))
        sc = metrics_ref.score(lab + 1, lab + 1, gt)
        row += [f"hash{salt}: groups {len(g)} ARI {ncuts_ref.adjusted_rand_index(lab, ref):.5f} dAP {sc['ap']-meta['scores']['ap']:+.4f} dS {sc['S_assoc']-meta['scores']['S_assoc']:+.4f}"]
    print(*row, flush=True)
