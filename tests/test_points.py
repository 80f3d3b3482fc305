"""Next rows: TARL radius-mean pooling and 1-NN re-projection."""
import numpy as np
import pytest

from oracle import points_ref


def _scene(seed, n_major=4000, per_scan=30000, scans=4):
    from autoinst_amd import synth
    rng = np.random.default_rng(seed)
    major, _ = synth.surface_chunk(n_major, seed=seed, extent=16.0)
    # "scan" points: jittered copies of surface points, several per major voxel, some voxels empty
    base, _ = synth.surface_chunk(n_major, seed=seed, extent=16.0)
    keep = rng.random(base.shape[0]) > 0.1
    src = np.concatenate([base[keep][rng.integers(0, keep.sum(), per_scan)] + rng.normal(0, 0.08, (per_scan, 3)) for _ in range(scans)])
    feat = rng.standard_normal((src.shape[0], 96)).astype(np.float32)
    return major, src, feat


def test_oracle_pooling_against_brute_force():
    major, src, feat = _scene(1, n_major=300, per_scan=2000, scans=2)
    got = points_ref.tarl_pool(major, src, feat, 0.175)
    d = np.linalg.norm(major[:, None, :] - src[None, :, :], axis=2)
    for i in range(major.shape[0]):
        idx = np.flatnonzero(d[i] < 0.175)
        exp = feat[idx].astype(np.float64).mean(0) if idx.size else np.zeros(96)
        assert np.abs(got[i] - exp).max() <= 1e-14


def test_oracle_nn1_against_brute_force():
    rng = np.random.default_rng(2)
    a, b = rng.random((500, 3)) * 5, rng.random((80, 3)) * 5
    lab = rng.integers(0, 9, (80, 3)).astype(np.float64)
    got = points_ref.nn1_reproject(np.zeros((500, 3)), a, lab, b)
    idx = np.linalg.norm(a[:, None] - b[None], axis=2).argmin(1)
    assert np.array_equal(got, lab[idx])


@pytest.mark.gpu
def test_tarl_pool_matches_oracle():
    from autoinst_amd import points_api
    major, src, feat = _scene(3)
    got = points_api.tarl_pool(major, src, feat)
    exp = points_ref.tarl_pool(major, src, feat, 0.175)
    assert got.shape == exp.shape == (major.shape[0], 96)
    assert np.array_equal(~got.any(1), ~exp.any(1)), "zero rows (no feature in radius) differ"
    assert np.abs(got - exp).max() <= 1e-12
    assert (~exp.any(1)).sum() > 0, "fixture must contain points without a feature"


@pytest.mark.gpu
def test_tarl_pool_edge_cases():
    from autoinst_amd import points_api
    q = np.array([[0.0, 0, 0], [10.0, 0, 0], [0.1, 0, 0]])
    s = np.array([[0.05, 0, 0], [0.2, 0, 0]])
    f = np.array([[1.0, 2.0], [3.0, 6.0]], dtype=np.float32)
    out = points_api.tarl_pool(q, s, f, radius=0.175)
    assert np.array_equal(out, np.array([[1.0, 2.0], [0.0, 0.0], [2.0, 4.0]]))
    assert np.array_equal(points_api.tarl_pool(q, np.zeros((0, 3)), np.zeros((0, 2), np.float32)), np.zeros((3, 2)))


@pytest.mark.gpu
def test_nn1_reprojection_matches_oracle():
    from autoinst_amd import points_api, synth
    major, _ = synth.surface_chunk(5000, seed=5, extent=18.0)
    rng = np.random.default_rng(5)
    fine = major[rng.integers(0, major.shape[0], 60000)] + rng.normal(0, 0.1, (60000, 3))
    fine = np.concatenate([fine, rng.uniform(-30, 30, (200, 3))])  # some far outside the cloud
    colors = rng.random((major.shape[0], 3))
    got = points_api.nn1_reproject(np.zeros((fine.shape[0], 3)), fine, colors, major)
    exp = points_ref.nn1_reproject(np.zeros((fine.shape[0], 3)), fine, colors, major)
    assert np.array_equal(got, exp)
    got_r = points_api.nn1_reproject(np.zeros((fine.shape[0], 3)), fine, colors, major, max_radius=0.2)
    exp_r = points_ref.nn1_reproject(np.zeros((fine.shape[0], 3)), fine, colors, major, max_radius=0.2)
    assert np.array_equal(got_r, exp_r)


@pytest.mark.gpu
def test_tarl_file_golden_through_pooling_and_affinity():
    """On-disk format -> hot path, on the device: tests/golden/formats/tarl/000042.bin (the bytes the reference's reader
    `get_tarl_features`, kitti_odometry_dataset.py:251-281, was run on) is read with `formats.read_tarl_bin`, its 211 feature
    rows are pooled onto major-voxel points with `tarl_pool` (chunk_generation.py:243-256) and the pooled features go into
    `build_affinity` (ncuts_utils.py:135-149): pooling == oracle/points_ref to 1e-12, affinity == oracle/ncuts_ref to 1e-12,
    same pattern."""
    import os
    from conftest import GOLDEN
    from autoinst_amd import formats, ncuts_api as api, points_api
    from oracle import ncuts_ref
    d = os.path.join(GOLDEN, "formats")
    feat = formats.read_tarl_bin(os.path.join(d, "tarl", "000042.bin"))
    exp = np.load(os.path.join(d, "expected.npz"))["tarl_000042"]
    assert feat.dtype == np.float32 and np.array_equal(feat, exp)      # what the reference's reader returned
    rng = np.random.default_rng(42)
    src = rng.uniform(0.0, 3.0, (feat.shape[0], 3))                    # the scan points the 211 feature rows belong to
    major = np.concatenate([src[rng.integers(0, src.shape[0], 150)] + rng.normal(0, 0.05, (150, 3)), rng.uniform(0, 3, (60, 3))])
    pooled = points_api.tarl_pool(major, src, feat)
    ref = points_ref.tarl_pool(major, src, feat, 0.175)
    assert np.array_equal(~pooled.any(1), ~ref.any(1)) and 0 < (~ref.any(1)).sum() < major.shape[0]
    assert np.abs(pooled - ref).max() <= 1e-12
    A = api.get_affinity_matrix(major, pooled, None, alpha=1.0, theta=0.5, gamma=0.0)
    B = ncuts_ref.affinity_sparse(major, ref, None, alpha=1.0, theta=0.5, gamma=0.0)
    assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
    assert (np.abs(A.data - B.data) / B.data).max() <= 1e-12
