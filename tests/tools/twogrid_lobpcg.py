"""CPU prototype: two-grid preconditioned LOBPCG over the 32-row task aggregates vs the device algorithm's
Lanczos step counts, on the largest connected segments of the 50k synthetic chunk."""
import sys, time, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as sla
import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
from autoinst_amd import synth
from oracle import ncuts_ref
from scipy.sparse.csgraph import connected_components
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
ch = synth.synthetic_chunk(n, 0, tarl=True)
A = ncuts_ref.affinity_sparse(ch["points"], ch["tarl"], None, alpha=1.0, theta=0.5, gamma=0.0).tocsr()
nc, lab = connected_components(A, directed=False)
sizes = np.bincount(lab)
order = np.argsort(-sizes)
# Morton-ish order: sort points by cell key so that consecutive rows are neighbours
P = ch["points"]; cell = np.floor((P - P.min(0)) / 1.0).astype(np.int64)
def part(x):
    x = x & 0x3ff; x = (x ^ (x << 16)) & 0xff0000ff; x = (x ^ (x << 8)) & 0x0300f00f; x = (x ^ (x << 4)) & 0x030c30c3; x = (x ^ (x << 2)) & 0x09249249; return x
key = part(cell[:,0]) | (part(cell[:,1]) << 1) | (part(cell[:,2]) << 2)
for ci in order[:4]:
    rows = np.where(lab == ci)[0]
    rows = rows[np.argsort(key[rows], kind="stable")]
    W = A[rows][:, rows].tocsr(); m = W.shape[0]
    if m < 2000: continue
    import gpu_model
    _res = gpu_model.lanczos_fiedler(W, rows.astype(np.int64))
    print('   model Lanczos:', [type(x).__name__ if hasattr(x,'shape') else x for x in _res], flush=True)
    d = np.asarray(W.sum(1)).ravel(); s = 1/np.sqrt(d)
    M = sp.diags(s) @ W @ sp.diags(s)
    L = sp.identity(m) - M
    u1 = np.sqrt(d / d.sum())
    # reference: plain Lanczos steps (model) -- count matvecs of scipy eigsh without shift-invert as a proxy
    cnt = [0]
    def mv(x): cnt[0] += 1; return M @ x - u1 * (u1 @ x)
    t0 = time.time(); vals, vecs = sla.eigsh(sla.LinearOperator((m, m), matvec=mv, dtype=np.float64), k=1, which="LA", tol=1e-10, ncv=min(m - 1, 400), maxiter=20)
    lanczos_mv = cnt[0]
    lam2 = 1 - vals[0]
    # aggregates: 32 consecutive rows
    G = 32; na = (m + G - 1) // G
    agg = np.arange(m) // G
    Pm = sp.csr_matrix((np.sqrt(d), (np.arange(m), agg)), shape=(m, na))   # D^{1/2} piecewise constant: null vector representable
    Pn = np.sqrt(np.asarray(Pm.multiply(Pm).sum(0)).ravel()); Pm = Pm @ sp.diags(1 / Pn)
    Ac = (Pm.T @ L @ Pm).toarray()
    Acp = np.linalg.pinv(Ac, rcond=1e-10)
    om = 0.7
    lcnt = [0]
    def prec(r):
        # two-grid, symmetric: pre-smooth, coarse correct, post-smooth (smoother: damped Richardson, diag(L) ~ 1)
        r = r - u1[:, None] * (u1 @ r) if r.ndim == 2 else r - u1 * (u1 @ r)
        x = om * r
        res = r - L @ x; lcnt[0] += 1
        x = x + Pm @ (Acp @ (Pm.T @ res))
        res = r - L @ x; lcnt[0] += 1
        x = x + om * res
        return x
    for label, Mop in (("two-grid", sla.LinearOperator((m, m), matvec=prec, matmat=prec, dtype=np.float64)), ("none", None)):
        rng = np.random.default_rng(0)
        X = rng.standard_normal((m, 1))
        hist = []
        lcnt[0] = 0
        try:
            w, v, rn = sla.lobpcg(L, X, M=Mop, Y=u1[:, None], tol=1e-10, maxiter=400, largest=False, retResidualNormsHistory=True, verbosityLevel=0)[:3] if False else (None, None, None)
        except Exception as e:
            pass
        res = sla.lobpcg(L, X, M=Mop, Y=u1[:, None], tol=1e-10, maxiter=400, largest=False, retResidualNormsHistory=True)
        w, v, hist = res
        it = len(hist)
        print(f"segment rows {m} aggregates {na} lam2 {lam2:.3e} | lanczos(ARPACK restarted) matvecs {lanczos_mv} | lobpcg[{label}] iterations {it} lam {w[0]:.3e} final resid {float(np.ravel(hist[-1])[0]):.2e} prec L-matvecs {lcnt[0]}", flush=True)
