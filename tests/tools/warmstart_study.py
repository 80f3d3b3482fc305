"""CPU study (round 5, verdict item 2): does a child's Lanczos solve get shorter when it starts from the parent's HIGHER Ritz vectors?

Round 2 tried the parent's Fiedler vector restricted to the child (8 % fewer steps): inside a child that vector is nearly the null
vector that is projected out.  The candidates here are the parent's 3rd .. p-th Ritz vectors (rank 1 .. p - 2 of its T_m; rank 0 is the
Fiedler vector that was cut): restricted to the child's rows, summed with their Ritz values as weights, the child's u1 projected
out.  The parent's Krylov basis is all still in HBM on the device (every vector is kept), so each extra Ritz vector costs one more
fk_ritz pass.  Children of a component split have no solved parent: they inherit the restricted vectors of the last SOLVED ancestor
(`--inherit`, default) or fall back to the hash vector (`--no-inherit`).

The whole recursion of the device algorithm (tests/gpu_model.py: Lanczos without re-orthogonalisation on M = I - L_sym with u1
projected out, component split, the reference's sweep) runs once per variant; convergence is tested at EVERY step (the device's
residual-trend schedule lands within a few steps of it), so `steps` is the exact number of steps to |beta_m s_m| <= 1e-10.

    python tests/tools/warmstart_study.py N MODE SEED [variant ...]   variants: hash  p3  p4  p6  p3f (incl. the Fiedler vector) ...
        -> one JSON line per variant: solves, steps, row_steps (sum of rows x steps), groups, partition equal to the hash variant's
"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
from scipy.linalg import eigh_tridiagonal
from scipy.sparse.csgraph import connected_components

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ncuts_ref  # noqa: E402
from oracle.gen_fullsize import MODES, chunk_for  # noqa: E402
import gpu_model as gm  # noqa: E402


def rr_start(w, cols):
    """Best start vector in the span of `cols` (n x k): Rayleigh-Ritz of M = D^-1/2 (w + I) D^-1/2 with u1 projected out.  Costs k SpMVs."""
    n = w.shape[0]
    d = np.asarray(w.sum(axis=0)).ravel() + 1.0
    s = 1.0 / np.sqrt(d)
    Wm = (sp.diags(s) @ (w + sp.identity(n)) @ sp.diags(s)).tocsr()
    u1 = np.sqrt(d / d.sum())
    Q = cols - np.outer(u1, u1 @ cols)
    Q, R = np.linalg.qr(Q)
    keep = np.abs(np.diag(R)) > 1e-10 * np.abs(np.diag(R)).max()
    Q = Q[:, keep]
    if Q.shape[1] == 0:
        return None
    H = Q.T @ (Wm @ Q)
    wv, Z = np.linalg.eigh(0.5 * (H + H.T))
    return Q @ Z[:, -1]


def rr_start_device_form(w, cols, ids, eps=float(os.environ.get("WS_RR_EPS", "1e-3"))):
    """`rr_start` as a device would form it: ONE pass over the child's entries gives Z = M Q (an SpMM with K vectors) and the raw sums
    a = Q^T u1, G = Q^T Q, H = Q^T Z; the projection of u1 is algebra (M u1 = u1: G~ = G - a a^T, H~ = H - a a^T); the K x K generalised
    problem H~ c = theta G~ c is solved after dropping the directions of G~ below 1e-10 of its largest eigenvalue; the start vector is
    sum_k c_k q_k - (c . a) u1 + eps * hash."""
    n = w.shape[0]
    d = np.asarray(w.sum(axis=0)).ravel() + 1.0
    s = 1.0 / np.sqrt(d)
    Wm = (sp.diags(s) @ (w + sp.identity(n)) @ sp.diags(s)).tocsr()
    u1 = np.sqrt(d / d.sum())
    Q = cols * np.sqrt(n)          # entries of order 1
    a = Q.T @ u1
    if os.environ.get("WS_RR_TWO_PASS") == "1":
        # two passes: the u1 components first, then everything on the PROJECTED vectors (no cancellation in G~ / H~)
        Qp = Q - np.outer(u1, a)
        Z = Wm @ Qp
        G = Qp.T @ Qp
        H = Qp.T @ Z
    else:
        Z = Wm @ Q
        G = Q.T @ Q - np.outer(a, a)
        H = Q.T @ Z - np.outer(a, a)
    G = 0.5 * (G + G.T)
    H = 0.5 * (H + H.T)
    lam, U = np.linalg.eigh(G)
    keep = lam > 1e-10 * lam.max()
    if not keep.any():
        return None
    T = U[:, keep] / np.sqrt(lam[keep])      # G~-orthonormal basis
    th, Y = np.linalg.eigh(T.T @ H @ T)
    c = T @ Y[:, -1]
    return (Q @ c - (c @ a) * u1) * np.sqrt(n) + eps * gm.start_vector(ids)   # (Q c has unit norm: entries of order 1 before the hash admixture)


def lanczos(w, ids, start=None, tol=1e-10, max_iter=4000, n_extra=0):
    """gm.lanczos_fiedler with a given start vector; also returns `n_extra` further Ritz vectors (ranks 1 ..) with their Ritz values."""
    n = w.shape[0]
    d = np.asarray(w.sum(axis=0)).ravel() + 1.0
    s = 1.0 / np.sqrt(d)
    Wm = (sp.diags(s) @ (w + sp.identity(n)) @ sp.diags(s)).tocsr()
    u1 = np.sqrt(d / d.sum())
    v = gm.start_vector(ids) if start is None else np.array(start, dtype=np.float64)
    v -= u1 * (u1 @ v)
    nv = np.linalg.norm(v)
    if not (nv > 1e-12 * np.sqrt(n)):   # the warm vector was (numerically) u1: no information
        v = gm.start_vector(ids)
        v -= u1 * (u1 @ v)
        nv = np.linalg.norm(v)
    v /= nv
    V = [v]
    alpha, beta = [], []
    v_prev, b_prev = np.zeros(n), 0.0
    m_cap = min(max_iter, n - 1)
    resid = np.inf
    for j in range(m_cap):
        y = Wm @ v
        a = v @ y
        wv = y - a * v - b_prev * v_prev
        g = u1 @ wv
        b = np.sqrt(max(wv @ wv - g * g, 0.0))
        wv = wv - g * u1
        alpha.append(a)
        beta.append(b)
        m = j + 1
        done = (b <= 1e-14) or (m == m_cap)
        th, S = eigh_tridiagonal(np.array(alpha), np.array(beta[:-1]), select="i", select_range=(m - 1, m - 1))
        resid = abs(b * S[-1, 0])
        if resid <= tol or done:
            break
        v_prev, b_prev = v, b
        v = wv / b
        V.append(v)
    m = len(alpha)
    k = min(1 + n_extra, m)
    th, S = eigh_tridiagonal(np.array(alpha), np.array(beta[:-1]) if m > 1 else np.zeros(0), select="i", select_range=(m - k, m - 1))
    Y = np.zeros((n, k))
    for j in range(m):
        Y += np.outer(V[j], S[j, :])
    Y /= np.linalg.norm(Y, axis=0)
    ev = gm.fix_sign(Y[:, -1].copy())
    extra = [(th[k - 1 - r], gm.fix_sign(Y[:, k - 1 - r].copy())) for r in range(1, k)]   # rank 1, 2, ...
    return ev, m, resid, d, (th[-1], ev), extra


def run_variant(A, n, T, variant, inherit=True):
    """The whole recursion with one start-vector rule.  Returns stats and canonical labels."""
    hashv = variant == "hash"
    dev = variant.startswith("dev")   # devP: like pP, but what the device would form: (sum of the Ritz vectors)|child * sqrt(n_parent) + 0.01 * hash
    if dev:
        variant = "p" + variant[3:]
    devrr = variant.startswith("rrdev")   # rrdevK: the Rayleigh-Ritz start in the form a device would compute (rr_start_device_form); K products counted as 3 steps (one SpMM pass)
    if devrr:
        variant = "rr" + variant[5:]
    free = variant.endswith("free")   # rrKfree: the K extra products of the Rayleigh-Ritz start are NOT counted (the ceiling of that form)
    if free:
        variant = variant[:-4]
    rr = variant.startswith("rr")    # rrK: Rayleigh-Ritz in the child over the parent's K top Ritz vectors (Fiedler included), K extra SpMVs
    with_f = variant.endswith("f") or rr
    p = 0 if hashv else int(variant[2:]) + 1 if rr else int(variant[1:].rstrip("f"))
    n_extra = 0 if hashv else max(p - 2, 0)
    st = {"variant": (("dev" + variant[1:]) if dev else variant) + ("free" if free else "") + ("(device form)" if devrr else ""), "solves": 0, "steps": 0, "row_steps": 0, "warm_solves": 0, "per_solve": []}
    groups = []

    def rec(w, lab, warm):
        nn = w.shape[0]
        if not gm._eligible(nn, n, 0.01):
            groups.append(lab)
            return
        ncomp, comp = connected_components(w, directed=False)
        if ncomp > 1:
            if not (0.0 < T):
                groups.append(lab)
                return
            for idx in gm.split_components(ncomp, comp):
                rec(w[idx][:, idx], lab[idx], (warm[idx] if (warm is not None and inherit) else None))
            return
        extra_steps = 0
        if rr and warm is not None:
            extra_steps = 3 if devrr else warm.shape[1]
            warm = rr_start_device_form(w, warm, lab) if devrr else rr_start(w, warm)
        ev, m, resid, d, fied, extra = lanczos(w, lab, start=warm, n_extra=n_extra)
        st["solves"] += 1
        if free:
            extra_steps = 0
        st["steps"] += m + extra_steps
        st["row_steps"] += nn * (m + extra_steps)
        st["warm_solves"] += warm is not None
        st["per_solve"].append((nn, m, warm is not None))
        mask, mcut, _ = gm.sweep(ev, d, w)
        if not (mcut < T):
            groups.append(lab)
            return
        wv = None
        if rr:
            wv = np.stack([fied[1]] + [y for _, y in extra], 1)
        elif not hashv:
            wv = np.zeros(nn)
            for th, y in extra:
                wv += th * y
            if with_f:
                wv += fied[0] * fied[1]
            if not extra and not with_f:
                wv = None
        if dev and wv is not None:
            wv = wv * np.sqrt(nn) + float(os.environ.get("WS_EPS", "0.01")) * gm.start_vector(lab)
        rec(w[mask][:, mask], lab[mask], None if wv is None else wv[mask])
        rec(w[~mask][:, ~mask], lab[~mask], None if wv is None else wv[~mask])

    sys.setrecursionlimit(10000)
    rec(A, np.arange(n), None)
    labels = ncuts_ref.groups_to_labels(groups, n)
    st["groups"] = len(groups)
    return st, labels


def main(argv):
    n, mode, seed = int(argv[0]), argv[1], int(argv[2])
    rest = argv[3:]
    inherit = "--no-inherit" not in rest
    variants = [a for a in rest if not a.startswith("--")] or ["hash", "p3", "p4", "p6"]
    cfg = MODES[mode]
    ch = chunk_for(n, mode, seed)
    A = ncuts_ref.affinity_sparse(ch["points"], ch["tarl"], ch["dino"], alpha=cfg["alpha"], theta=cfg["theta"], gamma=cfg["gamma"])
    base = None
    for v in variants:
        t0 = time.time()
        st, lab = run_variant(A, n, cfg["T"], v, inherit)
        st["seconds"] = round(time.time() - t0, 1)
        st["n"], st["mode"], st["seed"], st["inherit"] = n, mode, seed, inherit
        if v == "hash":
            base = (st, lab)
        if base is not None:
            st["partition_equal_to_hash"] = bool(ncuts_ref.partitions_equal(lab, base[1]))
            st["row_steps_vs_hash"] = round(st["row_steps"] / base[0]["row_steps"], 4)
            st["steps_vs_hash"] = round(st["steps"] / base[0]["steps"], 4)
        ps = st.pop("per_solve")
        big = [(a, b) for a, b, wm in ps if a >= 2000]
        st["solves_ge_2000_rows"] = len(big)
        st["mean_steps_ge_2000_rows"] = round(float(np.mean([b for _, b in big])), 1) if big else None
        print(json.dumps(st), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
