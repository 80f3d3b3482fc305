"""CPU suite: the oracle against the golden vectors made with the imported reference."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN, ROOT, golden_names
from oracle import metrics_ref, ncuts_ref
import gpu_model


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    n = z["points"].shape[0]
    A = sp.csr_matrix((z["data"], z["indices"], z["indptr"]), shape=(n, n))
    tarl = z["tarl"].astype(np.float64) if z["tarl"].size else None
    dino = z["dino"].astype(np.float64) if z["dino"].size else None
    return z, A, tarl, dino


@pytest.mark.parametrize("name", golden_names())
def test_affinity_restatement_matches_golden(name):
    z, A, tarl, dino = load(name)
    B = ncuts_ref.affinity_sparse(z["points"], tarl, dino, alpha=float(z["alpha"]), theta=float(z["theta"]), gamma=float(z["gamma"]))
    assert np.array_equal(B.indptr, A.indptr) and np.array_equal(B.indices, A.indices)
    assert np.abs(B.data - A.data).max() <= 1e-15
    # reference facts: symmetric, unit diagonal, so remove_isolated_points is the identity
    assert abs(A - A.T).max() == 0.0
    assert np.all(A.diagonal() == 1.0)


@pytest.mark.parametrize("name", ["g1_blob_pair_spatial", "g3_tarl_spatial", "g4_trimodal"])
def test_dense_reference_arithmetic_equals_sparse(name):
    z, A, tarl, dino = load(name)
    D = ncuts_ref.affinity_dense(z["points"], tarl, dino, alpha=float(z["alpha"]), theta=float(z["theta"]), gamma=float(z["gamma"]))
    keep, D2 = ncuts_ref.remove_isolated_points(D)
    assert keep.all()
    assert np.abs(D2 - A.toarray()).max() <= 1e-15


@pytest.mark.parametrize("name", golden_names())
def test_top_level_quantities(name):
    z, A, _, _ = load(name)
    _, d = ncuts_ref.laplacian_sym(A)
    assert np.allclose(d, z["degree"], rtol=0, atol=1e-12)
    # costs of the golden Fiedler vector, by both cost routines
    ev = z["fiedler"]
    mask, mcut = ncuts_ref.get_min_ncut(ev, d, A, 10)
    maskf, mcutf = ncuts_ref.get_min_ncut(ev, d, A, 10, fast=True)
    assert np.array_equal(mask, z["top_mask"]) and mcut == pytest.approx(float(z["top_mcut"]), abs=1e-12)
    if int(z["n_components"]) == 1:
        assert np.array_equal(maskf, mask) and mcutf == pytest.approx(mcut, rel=1e-10)
        vals, ev2, _ = ncuts_ref.fiedler(A)
        assert np.sort(vals)[1] == pytest.approx(float(z["eigvals"][1]), rel=1e-8)
        assert np.abs(np.abs(ev2) - z["fiedler_abs"]).max() <= 1e-8


def _fresh_oracle(name, fast):
    code = (
        "import sys, numpy as np, scipy.sparse as sp\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from oracle import ncuts_ref\n"
        f"z = np.load({os.path.join(GOLDEN, name + '.npz')!r})\n"
        "n = z['points'].shape[0]\n"
        "A = sp.csr_matrix((z['data'], z['indices'], z['indptr']), shape=(n, n))\n"
        f"g = ncuts_ref.normalized_cut(A, n, np.arange(n), T=float(z['T']), split_lim=0.01, fast={fast})\n"
        "lab = ncuts_ref.groups_to_labels(g, n)\n"
        "print(' '.join(map(str, lab.tolist())))\n"
    )
    out = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True).stdout
    return np.array(out.split(), dtype=np.int64)


@pytest.mark.parametrize("name", golden_names())
def test_recursion_matches_reference_output(name):
    """Exact groups AND order, in a fresh interpreter (ARPACK's start-vector state is per process)."""
    z, _, _, _ = load(name)
    lab = _fresh_oracle(name, fast=False)
    assert np.array_equal(lab, z["labels"])


@pytest.mark.parametrize("name", ["g1_blob_pair_spatial", "g6_connected_tarl", "g6_connected_spatial"])
def test_recursion_in_process_connected(name):
    z, A, _, _ = load(name)
    n = A.shape[0]
    g = ncuts_ref.normalized_cut(A, n, np.arange(n), T=float(z["T"]), fast=True)
    assert ncuts_ref.partitions_equal(ncuts_ref.groups_to_labels(g, n), z["labels"])


@pytest.mark.parametrize("name", golden_names())
def test_device_algorithm_model_matches_reference(name):
    """The algorithm the HIP path implements (tests/gpu_model.py) reproduces the reference partition."""
    z, A, _, _ = load(name)
    n = A.shape[0]
    g = gpu_model.normalized_cut_model(A, n, np.arange(n), T=float(z["T"]))
    lab = ncuts_ref.groups_to_labels(g, n)
    assert (lab >= 0).all()
    assert ncuts_ref.partitions_equal(lab, z["labels"])


@pytest.mark.parametrize("name", ["scorer_a", "scorer_b"])
def test_scorer_matches_reference(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    got = metrics_ref.score(z["pred"], z["pred"], z["gt"])
    for k, v in got.items():
        assert v == pytest.approx(float(z["exp_" + k.replace(".", "_")]), abs=1e-12), k


def test_partition_helpers():
    a = np.array([3, 3, 1, 1, 2])
    b = np.array([0, 0, 5, 5, 9])
    assert ncuts_ref.partitions_equal(a, b)
    assert ncuts_ref.adjusted_rand_index(a, b) == pytest.approx(1.0)
    assert not ncuts_ref.partitions_equal(a, np.array([0, 1, 5, 5, 9]))


@pytest.mark.parametrize("n", [20_000, 200_000])
def test_connected_solves_of_model_and_scipy_lead_to_the_same_cuts(n):
    """The reference recursion with SciPy's own shift-invert eigsh on every CONNECTED segment (and the component
    split on disconnected ones, which is what the reference's recursion amounts to there: gpu_model.split_components)
    gives EXACTLY the partition (and group order) of the device algorithm's model, up to BASELINE.json's full
    200k-point chunk (the GPU suite shows device == model on the same graph): every connected solve leads to the
    same cut.  What is left between device and reference is the one <= 1 % remainder per disconnected segment that
    the reference does not split (tests/golden/full_*.npz, tests/tools/fullsize_delta.py)."""
    from autoinst_amd import synth
    T = 0.03
    ch = synth.synthetic_chunk(n, 0, tarl=True)
    A = ncuts_ref.affinity_sparse(ch["points"], ch["tarl"], alpha=1.0, theta=0.5, gamma=0.0)
    gh = gpu_model.normalized_cut_model(A, n, np.arange(n), T=T, connected_solver=lambda w: ncuts_ref.fiedler(w)[1])
    gm = gpu_model.normalized_cut_model(A, n, np.arange(n), T=T)
    assert len(gh) == len(gm) and all(np.array_equal(a, b) for a, b in zip(gh, gm))


def _sam_ids(n, views, seed):
    rng = np.random.default_rng(seed)
    ids = rng.integers(0, 4, (n, views))
    ids[rng.random((n, views)) < 0.3] = -1      # no SAM id in that view
    return ids


def _sam_golden_cases():
    z = np.load(os.path.join(GOLDEN, "sam_ref.npz"))
    for c in range(int(z["cases"])):
        yield {k: z[f"c{c}_{k}"] for k in ("points", "sam", "beta", "radius", "rows", "cols", "factor")}


def test_sam_restatement_equals_the_reference_function():
    """Row a5 pinned: tests/golden/sam_ref.npz holds the outputs of the reference's own `sam_label_distance`
    (image_utils.py:64-89, compiled from the reference file by oracle/gen_sam_golden.py); the oracle's literal
    restatement and its sparse form reproduce them exactly (one view, eight views, radius 0.6, a duplicate point, a pair on
    the radius, a never-seen point)."""
    from scipy.spatial.distance import cdist
    for g in _sam_golden_cases():
        pts, ids, beta, radius = g["points"], g["sam"], float(g["beta"]), float(g["radius"])
        ld, mask = ncuts_ref.sam_label_distance(ids, cdist(pts, pts), radius, beta)
        r, c = np.nonzero(mask)
        assert np.array_equal(r, g["rows"]) and np.array_equal(c, g["cols"])
        assert np.array_equal(ld[r, c], g["factor"])
        A = ncuts_ref.affinity_sparse(pts, None, None, alpha=0.0, theta=0.0, gamma=0.0, radius=radius, sam=ids, beta=beta).tocoo()
        ref = sp.csr_matrix((g["factor"], (g["rows"], g["cols"])), shape=A.shape).tocoo()
        assert np.array_equal(A.row, ref.row) and np.array_equal(A.col, ref.col)
        assert np.abs(A.data - ref.data).max() <= 1e-15


def test_sam_factor_sparse_equals_literal_dense():
    """Row a5 (beta != 0, never taken by the shipped configs): the sparse restatement of the SAM factor
    equals the literal double loop of image_utils.py:64-89 inside the reference's product order."""
    rng = np.random.default_rng(4)
    pts = rng.normal(0, 1.2, (160, 3))
    tarl = rng.normal(0, 1, (160, 96))
    tarl[::11] = 0.0
    sam = _sam_ids(160, 5, 9)
    kw = dict(alpha=1.0, theta=0.5, gamma=0.0, beta=0.7)
    D = ncuts_ref.affinity_dense(pts, tarl, None, sam=[sam], **kw)
    A = ncuts_ref.affinity_sparse(pts, tarl, None, sam=sam, **kw)
    assert np.array_equal(A.toarray() != 0, D != 0)
    assert np.abs(A.toarray() - D).max() <= 1e-15
    # the factor really acts: some pairs have differing ids in co-labelled views
    B = ncuts_ref.affinity_sparse(pts, tarl, None, alpha=1.0, theta=0.5, gamma=0.0)
    assert (A.data < B.data - 1e-6).any() and np.all(A.data <= B.data + 1e-15)
    with pytest.raises(ValueError):
        ncuts_ref.affinity_sparse(pts, tarl, None, alpha=1.0, theta=0.5, gamma=0.0, beta=0.7)


def test_random_small_clouds_model_equals_reference_with_eigsh():
    """Thirty random clouds: the model of the device algorithm == the reference recursion with SciPy's
    shift-invert eigsh on every connected segment (and the component split on disconnected ones),
    same groups in the same order."""
    rng = np.random.default_rng(99)
    bad = []
    for case in range(30):
        n = int(rng.integers(60, 1500))
        kind = case % 3
        if kind == 0:
            c = rng.normal(0, 4, (int(rng.integers(2, 6)), 3))
            pts = c[rng.integers(0, c.shape[0], n)] + rng.normal(0, rng.uniform(0.3, 1.0), (n, 3))
        elif kind == 1:
            xy = rng.uniform(-7, 7, (n, 2))
            pts = np.c_[xy, 0.3 * np.sin(xy[:, 0]) + rng.normal(0, 0.05, n)]
        else:
            pts = rng.uniform(-5, 5, (n, 3))
        T = float(rng.choice([0.02, 0.1, 0.3]))
        A = ncuts_ref.affinity_sparse(pts, None, alpha=1.0, theta=0.0, gamma=0.0)
        gh = gpu_model.normalized_cut_model(A, n, np.arange(n), T=T, connected_solver=lambda w: ncuts_ref.fiedler(w)[1])
        gm = gpu_model.normalized_cut_model(A, n, np.arange(n), T=T)
        if not (len(gh) == len(gm) and all(np.array_equal(a, b) for a, b in zip(gh, gm))):
            la, lb = ncuts_ref.groups_to_labels(gh, n), ncuts_ref.groups_to_labels(gm, n)
            bad.append((case, n, kind, T, len(gh), len(gm), float(ncuts_ref.adjusted_rand_index(la, lb))))
    assert not bad, bad


def test_two_cameras_sparse_equals_literal_dense():
    """The loops over cameras (ncuts_utils.py:118-123, :128-133): one SAM and one DINO factor per camera."""
    rng = np.random.default_rng(6)
    pts = rng.normal(0, 1.2, (140, 3))
    dino = [rng.normal(0, 1, (140, 384)), rng.normal(0, 1, (140, 384))]
    sam = [_sam_ids(140, 4, 1), _sam_ids(140, 3, 2)]
    kw = dict(alpha=1.0, theta=0.0, gamma=0.05, beta=0.9)
    D = ncuts_ref.affinity_dense(pts, None, dino, sam=sam, **kw)
    A = ncuts_ref.affinity_sparse(pts, None, dino, sam=sam, **kw)
    assert np.array_equal(A.toarray() != 0, D != 0) and np.abs(A.toarray() - D).max() <= 1e-15


def _full_pairs():
    import glob
    import re
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "full_*_*.npz"))):
        m = re.fullmatch(r"(full_\d+_[a-z]+_\d+)_([pw]\d+)", os.path.basename(p)[:-4])
        if m and os.path.exists(os.path.join(GOLDEN, m.group(1) + ".npz")):
            out.append((m.group(1), m.group(2)))
    return out


@pytest.mark.parametrize("base,tag", _full_pairs())
def test_the_oracle_disagrees_with_itself_only_under_reordering(base, tag):
    """The record behind the full-size tolerance (tests/test_gpu_goldens.py).  `_w*` = the same oracle run after
    other eigsh calls (another ARPACK start-vector state): the partition is the same up to a handful of points.
    `_p*` = the same points listed in another order: the <= 1 % remainders come out differently (SuperLU's
    elimination order and round-off decide which component eigsh returns), ARI 0.988-0.999 (0.988 on the 200k spatial
    chunk, where the device is at 0.994 from the same oracle run) -- at least as far from the oracle as the device is."""
    import json
    a = np.load(os.path.join(GOLDEN, base + ".npz"))
    b = np.load(os.path.join(GOLDEN, f"{base}_{tag}.npz"))
    la, lb = a["labels"].astype(np.int64), b["labels"].astype(np.int64)
    ari = ncuts_ref.adjusted_rand_index(la, lb)
    ma, mb = json.loads(str(a["meta"])), json.loads(str(b["meta"]))
    print(base, tag, "ARI", ari, "groups", ma["groups"], mb["groups"], {k: mb["scores"][k] - ma["scores"][k] for k in ("ap", "S_assoc", "p")})
    if tag.startswith("w"):
        assert ari >= 0.999999 and abs(ma["groups"] - mb["groups"]) <= 1
    else:
        assert 0.985 <= ari < 1.0 and abs(ma["groups"] - mb["groups"]) <= 3


@pytest.mark.parametrize("name", ["c1_10k_spatial", "c1_10k_tarl"])
def test_model_equals_the_imported_reference_at_cfg1_size(name):
    """configs[0]'s own size: the device algorithm's model gives the partition the reference module produced."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    tarl = z["tarl"].astype(np.float64) if z["tarl"].size else None
    n = z["points"].shape[0]
    A = ncuts_ref.affinity_sparse(z["points"], tarl, alpha=float(z["alpha"]), theta=float(z["theta"]), gamma=float(z["gamma"]))
    assert A.nnz == int(z["nnz"])
    g = gpu_model.normalized_cut_model(A, n, np.arange(n), T=float(z["T"]))
    assert len(g) == int(z["n_groups"]) and ncuts_ref.partitions_equal(ncuts_ref.groups_to_labels(g, n), z["labels"])
