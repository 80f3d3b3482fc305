"""ORACLE tooling -- generates ``tests/golden/*.npz`` in the BUILD container.

Run as ``python oracle/gen_golden.py`` where ``/root/reference`` is mounted.  It imports
the reference's own ``pipeline/ncuts/normalized_cut.py`` and
``pipeline/metrics/{metrics_class,modified_LSTQ}.py`` (pure NumPy / SciPy; they import
here), runs them on seeded inputs and stores inputs + expected outputs.
Only data is committed; no reference source travels.  The GPU box never runs this.

Every fixture also asserts, at generation time, that the restatement in
``oracle/ncuts_ref.py`` reproduces the imported reference exactly (same list of groups,
same order), which is what pins the oracle.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy
import scipy.sparse as sp
from scipy.sparse.csgraph import connected_components

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference/pipeline"

from autoinst_amd import synth  # noqa: E402
from oracle import metrics_ref, ncuts_ref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _import_reference_ncut():
    sys.path.insert(0, REF)
    from ncuts.normalized_cut import normalized_cut  # type: ignore
    return normalized_cut


def _walk_connectivity(w, n_orig, labels, T, split_lim=0.01):
    """Re-run the restated recursion, reporting whether any solved segment was disconnected."""
    out = {"solves": 0, "disconnected_solves": 0}

    def rec(w, labels, split_lim):
        if w.shape[0] > 2 and labels.shape[0] / (n_orig + 1e-8) > split_lim:
            out["solves"] += 1
            nc, _ = connected_components(w, directed=False)
            if nc > 1:
                out["disconnected_solves"] += 1
            _, ev, d = ncuts_ref.fiedler(w)
            mask, mcut = ncuts_ref.get_min_ncut(ev, d, w, 10, fast=True)
            if mcut < T:
                rec(w[mask][:, mask], labels[mask], 0.01)
                rec(w[~mask][:, ~mask], labels[~mask], 0.01)

    rec(w, labels, split_lim)
    return out


def _fresh(which, A, n, T):
    """Run one end-to-end normalized_cut in a fresh interpreter; returns the list of groups."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        sp.save_npz(os.path.join(td, "A.npz"), A)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", which, td, str(n), repr(T)], check=True)
        z = np.load(os.path.join(td, "out.npz"))
        return [z[f"g{k}"] for k in range(int(z["count"]))]


def _worker(which, td, n, T):
    A = sp.load_npz(os.path.join(td, "A.npz"))
    if which == "ref":
        groups = _import_reference_ncut()(A, n, np.arange(n), T=T, split_lim=0.01)
    else:
        groups = ncuts_ref.normalized_cut(A, n, np.arange(n), T=T, split_lim=0.01, fast=(which == "oracle_fast"))
    np.savez(os.path.join(td, "out.npz"), count=len(groups), **{f"g{k}": g for k, g in enumerate(groups)})


def make_fixture(name, points, tarl, dino, cfg, ref_ncut, check_dense=True):
    alpha, theta, gamma, T = cfg["alpha"], cfg["theta"], cfg["gamma"], cfg["T"]
    n = points.shape[0]
    A = ncuts_ref.affinity_sparse(points, tarl, dino, alpha=alpha, theta=theta, gamma=gamma)
    if check_dense:
        Ad = ncuts_ref.affinity_dense(points, tarl, dino, alpha=alpha, theta=theta, gamma=gamma)
        keep, _ = ncuts_ref.remove_isolated_points(Ad)
        assert keep.all(), "A_ii = 1 => no isolated rows"
        Ad = sp.csr_matrix(Ad)
        Ad.sort_indices()
        assert np.array_equal(Ad.indptr, A.indptr) and np.array_equal(Ad.indices, A.indices), name
        err = np.abs(Ad.data - A.data).max()
        assert err <= 1e-15, (name, err)
    # top level quantities, by the same calls the reference makes
    eigvals, ev, d = ncuts_ref.fiedler(A)
    mn, mx = ev.min(), ev.max()
    costs = np.full(10, np.nan)
    if not np.allclose(mn, mx):
        with np.errstate(divide="ignore", invalid="ignore"):
            for k, t in enumerate(np.linspace(mn, mx, 10, endpoint=False)):
                costs[k] = ncuts_ref.ncut_cost(A, d, ev > t)
    mask, mcut = ncuts_ref.get_min_ncut(ev, d, A, 10)
    # the imported reference end to end.  ARPACK keeps its start-vector RNG state between
    # eigsh calls of one process, and on disconnected segments the returned null-space
    # vector depends on it, so the reference and each restatement run in FRESH processes
    # (the state a user of run_pipeline.py would see for the first chunk).
    groups_ref = _fresh("ref", A, n, T)
    groups_or = _fresh("oracle", A, n, T)
    groups_fast = _fresh("oracle_fast", A, n, T)
    assert len(groups_ref) == len(groups_or) and all(np.array_equal(a, b) for a, b in zip(groups_ref, groups_or)), name
    lab = ncuts_ref.groups_to_labels(groups_ref, n)
    assert (lab >= 0).all()
    fast_equal = ncuts_ref.partitions_equal(lab, ncuts_ref.groups_to_labels(groups_fast, n))
    nc, _ = connected_components(A, directed=False)
    walk = _walk_connectivity(A, n, np.arange(n), T)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        points=points,
        tarl=np.zeros((0, 0), np.float32) if tarl is None else tarl.astype(np.float32),
        dino=np.zeros((0, 0), np.float32) if dino is None else dino.astype(np.float32),
        alpha=alpha, theta=theta, gamma=gamma, T=T, split_lim=0.01, radius=1.0,
        indptr=A.indptr.astype(np.int64), indices=A.indices.astype(np.int32), data=A.data,
        degree=d, eigvals=np.sort(eigvals), fiedler_abs=np.abs(ev), fiedler=ev,
        costs=costs, top_mask=mask, top_mcut=mcut,
        labels=lab, n_groups=len(groups_ref), group_sizes=np.array([len(g) for g in groups_ref]),
        n_components=nc, solves=walk["solves"], disconnected_solves=walk["disconnected_solves"],
        fast_equal=fast_equal, versions=np.array([np.__version__, scipy.__version__]),
    )
    print(f"{name}: n={n} nnz={A.nnz} comps={nc} groups={len(groups_ref)} solves={walk['solves']} "
          f"disconnected_solves={walk['disconnected_solves']} fast_equal={fast_equal} eig={np.sort(eigvals)} mcut={mcut:.4g}")


def make_cfg1_fixture(name, n, seed, extent, cfg, with_tarl):
    """A fixture at configs[0]'s own size (N = 10k, extent 25 m): the IMPORTED reference
    end to end (``normalized_cut.py:37-63``, dense ``D.todense()`` and all: ~2 min and a few GB here), the
    restatement asserted equal to it, and the dense affinity lines (``ncuts_utils.py:60-67,135-156``) asserted
    equal to the sparse restatement.  Stored: inputs, ground truth, the reference's labels and group sizes,
    how many of its solves met a disconnected segment, and its scores."""
    import time
    points, gt = synth.surface_chunk(n, seed=seed, extent=extent)
    tarl = synth.surrogate_features(gt, 96, seed) if with_tarl else None
    alpha, theta, gamma, T = cfg["alpha"], cfg["theta"], cfg["gamma"], cfg["T"]
    A = ncuts_ref.affinity_sparse(points, tarl, None, alpha=alpha, theta=theta, gamma=gamma)
    Ad = ncuts_ref.affinity_dense(points, tarl, None, alpha=alpha, theta=theta, gamma=gamma)
    keep, _ = ncuts_ref.remove_isolated_points(Ad)
    assert keep.all()
    Ad = sp.csr_matrix(Ad)
    Ad.sort_indices()
    assert np.array_equal(Ad.indptr, A.indptr) and np.array_equal(Ad.indices, A.indices), name
    assert np.abs(Ad.data - A.data).max() <= 1e-15, name
    del Ad
    t0 = time.perf_counter()
    groups_ref = _fresh("ref", A, n, T)
    t_ref = time.perf_counter() - t0
    groups_fast = _fresh("oracle_fast", A, n, T)
    assert len(groups_ref) == len(groups_fast) and all(np.array_equal(a, b) for a, b in zip(groups_ref, groups_fast)), name
    lab = ncuts_ref.groups_to_labels(groups_ref, n)
    assert (lab >= 0).all()
    walk = _walk_connectivity(A, n, np.arange(n), T)
    nc, _ = connected_components(A, directed=False)
    gts = np.array(gt, copy=True)
    gts[::97] = 0
    cl = ncuts_ref.canonical_labels(lab) + 1
    sc = metrics_ref.score(cl, cl, gts)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), points=points, gt=gt,
        tarl=np.zeros((0, 0), np.float32) if tarl is None else tarl.astype(np.float32),
        alpha=alpha, theta=theta, gamma=gamma, T=T, split_lim=0.01, radius=1.0, nnz=A.nnz,
        labels=lab.astype(np.int32), n_groups=len(groups_ref), group_sizes=np.array([len(g) for g in groups_ref]),
        n_components=nc, solves=walk["solves"], disconnected_solves=walk["disconnected_solves"],
        reference_seconds=t_ref, versions=np.array([np.__version__, scipy.__version__]),
        **{"score_" + k.replace(".", "_"): v for k, v in sc.items()})
    print(f"{name}: n={n} nnz={A.nnz} comps={nc} groups={len(groups_ref)} solves={walk['solves']} "
          f"disconnected_solves={walk['disconnected_solves']} reference {t_ref:.0f} s scores={sc}")


def blob_pair(seed, n_each=150):
    """Two dense blobs joined by a thin bridge: one connected component, a clear Fiedler cut."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1.0, 1.0, size=(n_each, 3)) * np.array([1.2, 1.2, 0.6])
    b = rng.uniform(-1.0, 1.0, size=(n_each, 3)) * np.array([1.2, 1.2, 0.6]) + np.array([4.0, 0.0, 0.0])
    bridge = np.stack([np.linspace(1.0, 3.0, 5), np.zeros(5), np.zeros(5)], 1)
    return np.concatenate([a, b, bridge])


def largest_component(points, *feats):
    A = ncuts_ref.affinity_sparse(points, alpha=1.0)
    _, cc = connected_components(A, directed=False)
    keep = cc == np.bincount(cc).argmax()
    return (points[keep],) + tuple(None if f is None else f[keep] for f in feats) + (keep,)


def scorer_fixture(name, seed, n, RefMetrics, n_gt=12, n_pred=15):
    rng = np.random.default_rng(seed)
    # blocky labels with noise so that IoUs spread over the AP thresholds
    gt = np.repeat(rng.integers(0, n_gt + 1, size=n // 50 + 1), 50)[:n]
    pred = gt.copy()
    flip = rng.random(n) < 0.25
    pred[flip] = rng.integers(0, n_pred + 1, size=int(flip.sum()))
    # a few tiny instances that filter_labels must send to background
    pred[:40] = n_pred + 5
    pred[0] = 0
    gt[1] = 0
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        m = RefMetrics("golden")
        out, aps = m.update_stats(pred.copy(), pred.copy(), gt.copy())
    finally:
        os.chdir(cwd)
    exp = {"p": out["precision"], "r": out["recall"], "f1": out["fScore"], "ap": aps["ap"],
           "ap0.25": aps["0.25"], "ap0.5": aps["0.5"], "S_assoc": aps["lstq"]}
    mine = metrics_ref.score(pred, pred, gt)
    for k in exp:
        assert abs(exp[k] - mine[k]) <= 1e-12, (name, k, exp[k], mine[k])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), pred=pred, gt=gt,
                        **{"exp_" + k.replace(".", "_"): v for k, v in exp.items()})
    print(name, exp)


def main_cfg1():
    """`python oracle/gen_golden.py --cfg1`: the two 10k-point fixtures only (minutes, a few GB)."""
    os.makedirs(OUT, exist_ok=True)
    make_cfg1_fixture("c1_10k_spatial", 10_000, 0, 25.0, dict(alpha=1.0, theta=0.0, gamma=0.0, T=0.075), False)
    make_cfg1_fixture("c1_10k_tarl", 10_000, 0, 25.0, dict(alpha=1.0, theta=0.5, gamma=0.0, T=0.03), True)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_ncut = _import_reference_ncut()
    spatial = dict(alpha=1.0, theta=0.0, gamma=0.0, T=0.075)
    tarl_sp = dict(alpha=1.0, theta=0.5, gamma=0.0, T=0.03)
    tri = dict(alpha=1.0, theta=0.5, gamma=0.1, T=0.005)

    # G1: connected blob pair, spatial only -> exact partition
    make_fixture("g1_blob_pair_spatial", blob_pair(1), None, None, spatial, ref_ncut)
    # G2: spatial, multi-component surface chunk
    p, gt = synth.surface_chunk(1000, seed=11, extent=14.0)
    make_fixture("g2_multicomp_spatial", p, None, None, spatial, ref_ncut)
    # G3: TARL + spatial with 5 % zero-feature rows
    p, gt = synth.surface_chunk(800, seed=12, extent=10.0)
    make_fixture("g3_tarl_spatial", p, synth.surrogate_features(gt, 96, 12), None, tarl_sp, ref_ncut)
    # G4: tri-modal with 384-d features incl. zero rows
    p, gt = synth.surface_chunk(400, seed=13, extent=7.0)
    make_fixture("g4_trimodal", p, synth.surrogate_features(gt, 96, 13), synth.surrogate_features(gt, 384, 14), tri, ref_ncut)
    # G5: 3k surface chunk at the three shipped thresholds (spatial / tarl+spatial / tri-modal T)
    p, gt = synth.surface_chunk(3000, seed=11, extent=14.0)
    f = synth.surrogate_features(gt, 96, 11)
    make_fixture("g5_surface3k_T0075", p, None, None, spatial, ref_ncut, check_dense=True)
    make_fixture("g5_surface3k_T003", p, f, None, tarl_sp, ref_ncut, check_dense=True)
    make_fixture("g5_surface3k_T0005", p, f, None, dict(tarl_sp, T=0.005), ref_ncut, check_dense=False)
    # G6: largest connected component of a surface chunk (connected at the top level)
    p, gt = synth.surface_chunk(2500, seed=21, extent=11.0)
    f = synth.surrogate_features(gt, 96, 21)
    pc, fc, keep = largest_component(p, f)
    make_fixture("g6_connected_tarl", pc, fc, None, tarl_sp, ref_ncut)
    make_fixture("g6_connected_spatial", pc, None, None, spatial, ref_ncut)

    # scorer goldens (reference Metrics needs CWD = pipeline/, MPLBACKEND=Agg)
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        from metrics.metrics_class import Metrics as RefMetrics  # type: ignore
    finally:
        os.chdir(cwd)
    scorer_fixture("scorer_a", 5, 20000, RefMetrics)
    scorer_fixture("scorer_b", 6, 9000, RefMetrics, n_gt=6, n_pred=9)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        _worker(sys.argv[2], sys.argv[3], int(sys.argv[4]), float(sys.argv[5]))
    elif len(sys.argv) > 1 and sys.argv[1] == "--cfg1":
        main_cfg1()
    else:
        main()
