"""Scorer and chunk merge: oracle checks on the CPU, device parity on the GPU."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import merge_ref, metrics_ref


# ----------------------------------------------------------------------------- helpers
def np_label_pairs(a, b, **_):
    """NumPy statement of what ai_label_pairs returns (test-side only)."""
    a, b = np.asarray(a, np.int64), np.asarray(b, np.int64)
    if a.size == 0:
        return np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int64)
    key = np.stack([a, b], 1)
    u, c = np.unique(key, axis=0, return_counts=True)
    return u[:, 0].astype(np.int32), u[:, 1].astype(np.int32), c.astype(np.int64)


def palette(k, seed):
    rng = np.random.default_rng(seed)
    return np.round(rng.random((k, 3)), 3) * 0.9 + 0.05


def overlapping_chunks(seed=0, n_obj=9, pts=400, street=300):
    """Three chunks along x that share objects in their overlaps: every chunk paints its own random
    colours, shared objects have bit-identical coordinates in both chunks (as map chunks do)."""
    rng = np.random.default_rng(seed)
    centres = np.stack([np.linspace(-30, 30, n_obj), rng.uniform(-6, 6, n_obj), rng.uniform(0, 1, n_obj)], 1)
    objs = [c + rng.normal(0, 0.6, (pts, 3)) for c in centres]
    ground = np.stack([rng.uniform(-35, 35, street * 3), rng.uniform(-8, 8, street * 3), np.zeros(street * 3)], 1)
    chunks = []
    for ci, (lo, hi) in enumerate([(-36, -8), (-18, 12), (4, 36)]):
        P, Cc = [], []
        pal = palette(n_obj, 100 + ci)
        for k, o in enumerate(objs):
            if lo <= centres[k, 0] <= hi:
                P.append(o)
                Cc.append(np.tile(pal[k], (o.shape[0], 1)))
        g = ground[(ground[:, 0] >= lo) & (ground[:, 0] <= hi)]
        P.append(g)
        Cc.append(np.zeros((g.shape[0], 3)))
        chunks.append((np.concatenate(P), np.concatenate(Cc)))
    return chunks


# ----------------------------------------------------------------------------- CPU: oracle + host arithmetic
def test_merge_oracle_unites_shared_instances():
    chunks = overlapping_chunks()
    P, Cc = merge_ref.merge_chunks_unite_instances2(chunks)
    # no duplicated coordinates left, every point of the inputs is present
    assert np.unique(P, axis=0).shape[0] == P.shape[0]
    allp = np.unique(np.concatenate([c[0] for c in chunks]), axis=0)
    assert P.shape[0] == allp.shape[0]
    # 9 objects + street: shared objects took the colour of the earlier chunk
    assert np.unique(Cc, axis=0).shape[0] == 10
    first_p, first_c = chunks[0]
    assert np.array_equal(P[: first_p.shape[0]], first_p) and np.array_equal(Cc[: first_c.shape[0]], first_c)


def test_merge_oracle_union_is_distinct_scalars():
    """The reference's `union` counts distinct scalar coordinates (np.unique without axis)."""
    a = (np.array([[0., 0., 0.], [1., 1., 1.], [2., 0., 1.]]), np.tile([0.2, 0.3, 0.4], (3, 1)))
    b = (np.array([[0.5, 0.5, 0.5], [1., 1., 1.], [3., 3., 3.]]), np.tile([0.7, 0.1, 0.1], (3, 1)))
    P, Cc = merge_ref.merge_chunks_unite_instances2([a, b])
    # box of a = [0,2]x[0,1]x[0,1]; 2 points of b inside; distinct scalars {0,1,2,0.5,3} = 5 -> iou 0.4 > 0.01
    assert P.shape[0] == 5
    assert np.array_equal(np.unique(Cc, axis=0), np.array([[0.2, 0.3, 0.4]]))


def test_host_scorer_arithmetic_matches_oracle(monkeypatch):
    """labels_api's arithmetic on a NumPy-made contingency table == the pinned oracle, bit for bit."""
    from autoinst_amd import labels_api
    monkeypatch.setattr(labels_api, "label_pairs", np_label_pairs)
    for name in ("scorer_a", "scorer_b"):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        got = labels_api.score(z["pred"], z["pred"], z["gt"])
        exp = metrics_ref.score(z["pred"], z["pred"], z["gt"])
        for k, v in exp.items():
            assert got[k] == v or (np.isnan(got[k]) and np.isnan(v)), (name, k)
            assert v == pytest.approx(float(z["exp_" + k.replace(".", "_")]), abs=1e-12)
    rng = np.random.default_rng(5)
    gt = rng.integers(0, 12, 30_000)
    pred = np.where(rng.random(30_000) < 0.8, gt * 3 % 14, rng.integers(0, 40, 30_000))
    alll = np.where(rng.random(30_000) < 0.5, pred, rng.integers(0, 60, 30_000))
    got, exp = labels_api.score(alll, pred, gt, 150), metrics_ref.score(alll, pred, gt, 150)
    for k, v in exp.items():
        assert got[k] == v, k


def test_metrics_object_accumulates_like_the_reference(monkeypatch):
    from autoinst_amd import labels_api
    monkeypatch.setattr(labels_api, "label_pairs", np_label_pairs)
    z = np.load(os.path.join(GOLDEN, "scorer_a.npz"))
    m = labels_api.Metrics("t", min_points=200)
    out, aps = m.update_stats(z["pred"].copy(), z["pred"].copy(), z["gt"])
    assert out["precision"] == pytest.approx(float(z["exp_p"]), abs=1e-12)
    assert out["recall"] == pytest.approx(float(z["exp_r"]), abs=1e-12)
    assert aps["ap"] == pytest.approx(float(z["exp_ap"]), abs=1e-12)
    assert aps["lstq"] == pytest.approx(float(z["exp_S_assoc"]), abs=1e-12)
    m.update_stats(z["pred"].copy(), z["pred"].copy(), z["gt"])
    assert len(m.sequence_metrics["ap"]) == 2 and m.sequence_metrics["p"][1] == pytest.approx(float(z["exp_p"]), abs=1e-12)


def test_color_ids_follow_np_unique_order():
    from autoinst_amd import labels_api
    rng = np.random.default_rng(1)
    pal = np.concatenate([np.zeros((1, 3)), palette(7, 3)])
    col = pal[rng.integers(0, 8, 500)]
    table, ids = labels_api._color_ids(col)
    assert np.array_equal(table, np.unique(col, axis=0)) and np.array_equal(table[ids], col)
    table, ids = labels_api._color_ids(col[np.any(col != 0, axis=1)])       # no black: id 0 stays reserved
    assert ids.min() == 1 and np.array_equal(table[1:], np.unique(col, axis=0)[1:])


# ----------------------------------------------------------------------------- GPU: parity
@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 777, 1_000_000])
def test_label_pairs_equal_numpy(ctx, n):
    from autoinst_amd import labels_api
    rng = np.random.default_rng(n)
    a = rng.integers(-1, 300, n).astype(np.int64)
    b = rng.integers(0, 50, n).astype(np.int64)
    got = labels_api.label_pairs(a, b, ctx=ctx)
    exp = np_label_pairs(a, b)
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)
    assert int(got[2].sum()) == n


@pytest.mark.gpu
def test_label_pairs_more_pairs_than_first_capacity(ctx):
    from autoinst_amd import labels_api
    n = 300_000
    a = np.arange(n, dtype=np.int32)[::-1].copy()
    b = (np.arange(n, dtype=np.int32) * 7) % 1000
    pa, pb, cnt = labels_api.label_pairs(a, b, ctx=ctx)
    assert pa.shape[0] == n and np.array_equal(pa, np.arange(n)) and np.all(cnt == 1)
    assert np.array_equal(pb, b[::-1])
    e = labels_api.label_pairs(np.zeros(0, np.int32), np.zeros(0, np.int32), ctx=ctx)
    assert all(x.shape[0] == 0 for x in e)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["scorer_a", "scorer_b"])
def test_device_score_equals_reference_golden(ctx, name):
    from autoinst_amd import labels_api
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    got = labels_api.score(z["pred"], z["pred"], z["gt"], ctx=ctx)
    exp = metrics_ref.score(z["pred"], z["pred"], z["gt"])
    for k, v in exp.items():
        assert got[k] == v, k
        assert got[k] == pytest.approx(float(z["exp_" + k.replace(".", "_")]), abs=1e-12)


@pytest.mark.gpu
def test_device_score_of_an_ncuts_result(ctx):
    """The scorer on the output of the hot path itself (20k chunk, synthetic ground truth)."""
    from autoinst_amd import labels_api, ncuts_api, synth
    ch = synth.synthetic_chunk(20_000, 11, tarl=True)
    g = ncuts_api.build_affinity(ch["points"], ch["tarl"], alpha=1.0, theta=0.5, gamma=0.0, ctx=ctx)
    labels, _, _ = ncuts_api.ncuts_labels(g, 20_000, 0.03)
    g.free()
    gt = ch["gt"].copy()
    gt[::97] = 0
    got = labels_api.score(labels + 1, labels + 1, gt, ctx=ctx)
    exp = metrics_ref.score(labels + 1, labels + 1, gt)
    for k, v in exp.items():
        assert got[k] == v, k


@pytest.mark.gpu
def test_unique_points_equal_oracle(ctx):
    from autoinst_amd import labels_api
    rng = np.random.default_rng(2)
    base = np.round(rng.normal(0, 5, (50_000, 3)), 1)          # many exact duplicates
    base[::1000] = 0.0
    base[500::1000, 0] = -0.0                                   # -0.0 == +0.0 is the same point
    keep = labels_api.unique_points(base, ctx=ctx)
    exp_p, _ = merge_ref.remove_duplicated_points(base, base)
    assert np.array_equal(base[keep], exp_p)
    assert np.all(np.diff(keep) > 0)
    one = labels_api.unique_points(np.zeros((1, 3)), ctx=ctx)
    assert np.array_equal(one, [0])


@pytest.mark.gpu
def test_merge_counts_equal_brute_force(ctx):
    from autoinst_amd import labels_api
    chunks = overlapping_chunks(seed=3)
    (mp, mc), (cp, cc) = chunks[0], chunks[1]
    t1, i1 = labels_api._color_ids(mc)
    t2, i2 = labels_api._color_ids(cc)
    center = cp.mean(0)
    r = labels_api.merge_associate(mp, i1, cp, i2, center, t1.shape[0], t2.shape[0], 40.0, ctx=ctx)
    lo, hi = center - 20.0, center + 20.0
    crop = np.all(mp >= lo, 1) & np.all(mp <= hi, 1)
    for id1 in range(1, t1.shape[0]):
        p1 = mp[crop & (i1 == id1)]
        assert r["n_points1"][id1] == p1.shape[0]
        assert r["n_scalars1"][id1] == np.unique(p1).shape[0]
        for id2 in range(1, t2.shape[0]):
            p2 = cp[i2 == id2]
            assert r["n_scalars2"][id2] == np.unique(p2).shape[0]
            inside = 0 if p1.shape[0] == 0 else int((np.all(p2 >= p1.min(0), 1) & np.all(p2 <= p1.max(0), 1)).sum())
            assert r["inter"][id1, id2] == inside
            union = np.unique(np.concatenate((p1, p2))).shape[0]
            assert r["n_scalars1"][id1] + r["n_scalars2"][id2] - r["common"][id1, id2] == union


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 7])
def test_merge_equals_oracle(ctx, seed):
    from autoinst_amd import labels_api
    chunks = overlapping_chunks(seed=seed)
    P, Cc = labels_api.merge_chunks_unite_instances2(chunks, ctx=ctx)
    eP, eC = merge_ref.merge_chunks_unite_instances2(chunks)
    assert np.array_equal(P, eP) and np.array_equal(Cc, eC)


def test_host_scorer_arithmetic_property(monkeypatch):
    """Random small label arrays (incl. no background, everything background, -1 labels, tiny min_points):
    the product's scorer arithmetic on a NumPy contingency table == the pinned oracle."""
    from hypothesis import given, settings, strategies as st
    from autoinst_amd import labels_api
    monkeypatch.setattr(labels_api, "label_pairs", np_label_pairs)

    @settings(max_examples=120, deadline=None)
    @given(st.integers(1, 60), st.integers(0, 2 ** 31 - 1), st.integers(1, 6), st.integers(1, 6), st.integers(1, 8), st.booleans())
    def check(n, seed, kp, kg, min_points, with_bg):
        rng = np.random.default_rng(seed)
        gt = rng.integers(0 if with_bg else 1, kg + 1, n)
        pred = rng.integers(-1 if seed % 3 == 0 else 0, kp + 1, n)
        alll = np.where(rng.random(n) < 0.5, pred, rng.integers(0, kp + 2, n))
        def run(f):
            try:
                return f(alll, pred, gt, min_points), None
            except ZeroDivisionError as e:     # the reference divides by tp + fn as well (metrics_class.py:229)
                return None, type(e)
        (got, ge), (exp, ee) = run(labels_api.score), run(metrics_ref.score)
        assert ge == ee
        if exp is None:
            return
        for k, v in exp.items():
            a, b = got[k], v
            assert (a == b) or (isinstance(a, float) and isinstance(b, float) and np.isnan(a) and np.isnan(b)), (k, a, b)

    check()


@pytest.mark.gpu
def test_merge_equals_oracle_on_random_small_maps(ctx):
    """Randomised small maps: coincident points between chunks, instances that are all street, shared
    coordinate values on a coarse grid (so the 'distinct scalars' union really overlaps), -0.0."""
    from autoinst_amd import labels_api
    rng = np.random.default_rng(77)
    for case in range(25):
        n_chunks = int(rng.integers(2, 4))
        base = np.round(rng.normal(0, 6, (int(rng.integers(30, 400)), 3)) * 4) / 4      # 0.25 m grid: many shared scalars
        base[rng.random(base.shape) < 0.02] *= -0.0
        chunks = []
        for c in range(n_chunks):
            sel = rng.random(base.shape[0]) < 0.7
            pts = base[sel] + (0.0 if rng.random() < 0.6 else np.round(rng.normal(0, 0.5, 3) * 4) / 4)
            k = int(rng.integers(1, 6))
            inst = rng.integers(0, k + 1, pts.shape[0])                                    # 0 = street
            pal = np.concatenate([np.zeros((1, 3)), palette(k, 1000 * case + c)])
            chunks.append((pts, pal[inst]))
        P, Cc = labels_api.merge_chunks_unite_instances2(chunks, ctx=ctx)
        eP, eC = merge_ref.merge_chunks_unite_instances2(chunks)
        assert np.array_equal(P, eP) and np.array_equal(Cc, eC), case
