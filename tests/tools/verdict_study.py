"""CPU study (round 3): at which Lanczos step is a segment's VERDICT decided?

For every connected solve of the device algorithm's recursion (tests/gpu_model.py) on one synthetic chunk this logs,
per convergence check (every 16 steps, as on the device): the top two Ritz values and their residuals, and whether the
sweep of the Ritz vector formed at that check already gives the FINAL mask (the one the 1e-10 vector gives).  From that:

  * leaf rule (a): first check with (1 - theta) - resid >= T   (normalized_cut.py:56: mcut >= lambda_2 always)
  * plain tolerance sweep: for tol in 1e-3 .. 1e-9, the step at which resid <= tol and whether the mask of THAT
    vector is the final mask (the margin of the 1e-10 setting)
  * certificate (b): first check at which `certify()` accepts, and whether it was right.

    python tests/tools/verdict_study.py [n] [mode] [seed]  -> one JSON line per solve + a summary line
"""
import json, os, sys, time
import numpy as np
import scipy.sparse as sp
from scipy.linalg import eigh_tridiagonal
from scipy.sparse.csgraph import connected_components
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ncuts_ref
from oracle.gen_fullsize import MODES, chunk_for
import gpu_model as gm

CHECK = 16
TOLS = [1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8, 1e-9]
KAPPA = 4.0      # |u_k|_inf <= KAPPA * max|ev|  (delocalised error; measured 0.5 .. 3.2)
SAFETY = 4.0


def certify(ev, d, coo, delta, T):
    """Is the sweep's verdict on `ev` stable under any perturbation of `ev` of size <= delta per entry?

    Returns (verdict, mask, mcut): verdict 'split' / 'leaf' when certified, None otherwise.  Entries within 2*delta of
    a threshold (the thresholds move with mn / mx by at most delta themselves) are uncertain for it; U_k = their degree
    volume.  Leaf: every threshold's cost, lowered as far as U_k allows, stays >= T.  Split: the winning mask has no
    uncertain entry, its (then exact) cost is < T, and no other threshold can get below it."""
    mn, mx = ev.min(), ev.max()
    n = ev.shape[0]
    if np.allclose(mn, mx):
        return None, np.zeros(n, bool), np.inf
    step = (mx - mn) / gm.NUM_CUTS
    K = gm.NUM_CUTS
    costs, lo, U = np.empty(K), np.empty(K), np.empty(K)
    masks = []
    vol = d.sum()
    for k in range(K):
        t = k * step + mn
        mask = ev > t
        masks.append(mask)
        unc = np.abs(ev - t) <= 2 * delta
        if k == 0:
            unc &= ev > mn      # the minimum itself is never above t_0
        U[k] = d[unc].sum()
        cut = coo.data[mask[coo.row] & ~mask[coo.col]].sum()
        a, b = d[mask].sum(), d[~mask].sum()
        with np.errstate(divide="ignore", invalid="ignore"):
            costs[k] = cut / a + cut / b
            cl = max(cut - U[k], 0.0)
            lo[k] = cl / min(a + U[k], vol) + cl / min(b + U[k], vol)
    ks = int(np.nanargmin(costs))
    if np.all(lo >= T):
        return "leaf", masks[ks], costs[ks]
    ok = U[ks] == 0.0 and costs[ks] < T
    for k in range(K):
        if k != ks and not (lo[k] > costs[ks]):
            ok = False
    return ("split" if ok else None), masks[ks], costs[ks]


def study_solve(w, ids, T, tol=1e-10):
    n = w.shape[0]
    d = np.asarray(w.sum(axis=0)).ravel() + 1.0
    s = 1.0 / np.sqrt(d)
    Wm = (sp.diags(s) @ (w + sp.identity(n)) @ sp.diags(s)).tocsr()
    u1 = np.sqrt(d / d.sum())
    v = gm.start_vector(ids)
    v -= u1 * (u1 @ v)
    v /= np.linalg.norm(v)
    V = [v]
    alpha, beta = [], []
    v_prev, b_prev = np.zeros(n), 0.0
    m_cap = min(4000, n - 1)
    ce = 1 if n <= 512 else CHECK
    checks = []
    for j in range(m_cap):
        y = Wm @ v
        a = v @ y
        wv = y - a * v - b_prev * v_prev
        g = u1 @ wv
        b = np.sqrt(max(wv @ wv - g * g, 0.0))
        wv = wv - g * u1
        alpha.append(a); beta.append(b)
        m = j + 1
        done = (b <= 1e-14) or (m == m_cap)
        if done or (m % ce == 0):
            lo_i = max(m - 2, 0)
            th, S = eigh_tridiagonal(np.array(alpha), np.array(beta[:-1]), select="i", select_range=(lo_i, m - 1))
            theta, sv = th[-1], S[:, -1]
            theta2 = th[0] if m > 1 else -1.0
            r2 = abs(b * S[-1, 0]) if m > 1 else 1.0
            resid = abs(b * sv[-1])
            checks.append((m, theta, theta2, resid, sv.copy(), r2))
            if resid <= tol or done:
                break
        v_prev, b_prev = v, b
        v = wv / b
        V.append(v)
    Vm = np.stack(V, 1)
    coo = w.tocoo()

    def ritz(c):
        x = Vm[:, :c[0]] @ c[4]
        return gm.fix_sign(x / np.linalg.norm(x))
    m_f, th_f, th2_f, r_f = checks[-1][:4]
    ev_f = ritz(checks[-1])
    mask_f, mcut_f, _ = gm.sweep(ev_f, d, w)
    split = bool(mcut_f < T)
    rec = {"n": n, "steps": m_f, "lam2": 1.0 - th_f, "mcut": float(mcut_f) if np.isfinite(mcut_f) else None, "split": split,
           "gap": th_f - th2_f}
    rec["leaf_fire"] = next((m for (m, th, _, r, _, _) in checks if (1.0 - th) - r >= T), None)
    rec["leaf_ok"] = (rec["leaf_fire"] is None) or (not split)
    # masks along the way
    info = []
    for c in checks:
        m, th, th2, r, sv, r2 = c
        if r > 2e-3 and m != m_f:
            continue
        ev = ritz(c)
        mk, mc, _ = gm.sweep(ev, d, w)
        same = bool(np.array_equal(mk, mask_f)) and (bool(mc < T) == split)
        gap = th - th2 - r2
        verdict = None
        if gap > 0 and r2 < 0.5 * (th - th2):
            delta = SAFETY * KAPPA * np.abs(ev).max() * (r / gap)
            verdict, mk2, mc2 = certify(ev, d, coo, delta, T)
            if verdict == "leaf":
                right = not split
            elif verdict == "split":
                right = split and bool(np.array_equal(mk2, mask_f))
            else:
                right = True
        else:
            right = True
        info.append({"m": m, "r": r, "same": same, "verdict": verdict, "right": right, "gap": gap, "r2": r2})
    rec["tol_sweep"] = {}
    for tl in TOLS:
        i = next((x for x in info if x["r"] <= tl), info[-1])
        rec["tol_sweep"]["%g" % tl] = (i["m"], i["same"])
    first_stable = m_f
    for x in reversed(info):
        if x["same"]:
            first_stable = x["m"]
        else:
            break
    rec["mask_stable_from"] = first_stable
    cf = next((x for x in info if x["verdict"]), None)
    rec["cert_fire"] = cf["m"] if cf else None
    rec["cert_resid"] = cf["r"] if cf else None
    rec["cert_verdict"] = cf["verdict"] if cf else None
    rec["cert_right"] = cf["right"] if cf else True
    rec["attempts"] = sum(1 for x in info if x["m"] <= (cf["m"] if cf else m_f) and x["gap"] > 0)
    rec["steps_cert"] = cf["m"] if cf else m_f
    rec["checks"] = [(x["m"], float("%.2e" % x["r"]), int(x["same"]), x["verdict"]) for x in info]
    return rec, mask_f, split


def run(n, mode, seed):
    cfg = MODES[mode]
    ch = chunk_for(n, mode, seed)
    A = ncuts_ref.affinity_sparse(ch["points"], ch["tarl"], ch["dino"], alpha=cfg["alpha"], theta=cfg["theta"], gamma=cfg["gamma"])
    T = cfg["T"]
    recs = []
    level = [(A, np.arange(n))]
    depth = 0
    lock = {"base": 0, "cert": 0, "stable": 0}
    while level:
        nxt = []
        lv = {"base": 0, "cert": 0, "stable": 0}
        for (w, lab) in level:
            nn = w.shape[0]
            if not gm._eligible(nn, n, 0.01):
                continue
            ncomp, comp = connected_components(w, directed=False)
            if ncomp > 1:
                for idx in gm.split_components(ncomp, comp):
                    nxt.append((w[idx][:, idx], lab[idx]))
                continue
            rec, mask, split = study_solve(w, lab, T)
            rec["depth"] = depth
            recs.append(rec)
            print(json.dumps(rec), flush=True)
            lv["base"] = max(lv["base"], rec["steps"]); lv["cert"] = max(lv["cert"], rec["steps_cert"])
            lv["stable"] = max(lv["stable"], rec["mask_stable_from"])
            if split:
                nxt.append((w[mask][:, mask], lab[mask]))
                nxt.append((w[~mask][:, ~mask], lab[~mask]))
        for k in lock:
            lock[k] += lv[k]
        level = nxt
        depth += 1
    rs = lambda f: int(sum(r["n"] * f(r) for r in recs))
    summ = {"summary": True, "n": n, "mode": mode, "seed": seed, "T": T, "solves": len(recs), "splits": sum(r["split"] for r in recs),
            "rowsteps_base": rs(lambda r: r["steps"]), "rowsteps_cert": rs(lambda r: r["steps_cert"]),
            "rowsteps_stable": rs(lambda r: r["mask_stable_from"]),
            "rowsteps_leaves_base": int(sum(r["n"] * r["steps"] for r in recs if not r["split"])),
            "lockstep": lock,
            "leaf_fired": sum(r["leaf_fire"] is not None for r in recs), "leaf_wrong": sum(not r["leaf_ok"] for r in recs),
            "cert_fired": sum(r["cert_fire"] is not None for r in recs), "cert_wrong": sum(not r["cert_right"] for r in recs),
            "attempts": sum(r["attempts"] for r in recs), "min_gap": min(r["gap"] for r in recs),
            "tol_sweep": {k: {"rowsteps": rs(lambda r: r["tol_sweep"][k][0]), "wrong_masks": sum(not r["tol_sweep"][k][1] for r in recs)}
                          for k in ("%g" % t for t in TOLS)}}
    print(json.dumps(summ), flush=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    mode = sys.argv[2] if len(sys.argv) > 2 else "tarl"
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    run(n, mode, seed)
