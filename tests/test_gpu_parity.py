"""GPU suite (-m gpu): the HIP path through the C ABI against goldens, the oracle and the model."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
from scipy.sparse.csgraph import connected_components

from conftest import GOLDEN, golden_names
from oracle import metrics_ref, ncuts_ref
import gpu_model

pytestmark = pytest.mark.gpu

CONNECTED = ["g1_blob_pair_spatial", "g6_connected_tarl", "g6_connected_spatial"]


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    n = z["points"].shape[0]
    A = sp.csr_matrix((z["data"], z["indices"], z["indptr"]), shape=(n, n))
    tarl = z["tarl"].astype(np.float64) if z["tarl"].size else None
    dino = z["dino"].astype(np.float64) if z["dino"].size else None
    return z, A, tarl, dino


@pytest.fixture(scope="module")
def api():
    from autoinst_amd import ncuts_api
    ncuts_api.default_context()
    return ncuts_api


@pytest.mark.parametrize("name", golden_names())
def test_affinity_matches_golden(api, name):
    z, A, tarl, dino = load(name)
    B = api.get_affinity_matrix(z["points"], tarl, dino, alpha=float(z["alpha"]), theta=float(z["theta"]), gamma=float(z["gamma"]))
    assert B.shape == A.shape and B.nnz == A.nnz
    assert np.array_equal(B.indptr, A.indptr), "row lengths differ"
    assert np.array_equal(B.indices, A.indices), "sparsity pattern differs"
    rel = np.abs(B.data - A.data) / A.data
    assert rel.max() <= 1e-12, rel.max()
    assert abs(B - B.T).max() == 0.0, "device affinity is not bitwise symmetric"
    assert np.all(B.diagonal() == 1.0)


@pytest.mark.parametrize("tdim,ddim", [(96, 0), (32, 48), (0, 64), (112, 16)])
def test_feature_factors_tiled_and_fallback_tiles_in_one_graph(api, tdim, ddim):
    """`k_weights_lanes`: a sparse sheet (tiles staged in LDS), a dense blob inside it (tiles with more than 256 distinct
    neighbours or more than 2048 entries take the per-entry fallback) and rows of more than 64 entries (second round of a
    tile) in ONE graph, feature widths that are multiples of 16 but not the reference's: values equal the oracle's to 1e-12,
    and (i, j) == (j, i) bit for bit whichever path computed them."""
    rng = np.random.default_rng(100 + tdim + ddim)
    sheet = np.c_[rng.uniform(-12, 12, (5000, 2)), rng.normal(0, 0.05, 5000)]
    blob = rng.normal(0, 0.35, (1500, 3)) + np.array([2.0, -3.0, 0.0])
    mid = rng.normal(0, 0.8, (1500, 3)) + np.array([-5.0, 4.0, 0.0])
    pts = np.concatenate([sheet, blob, mid])
    n = pts.shape[0]
    tarl = rng.normal(0, 0.4, (n, tdim)) if tdim else None
    if tarl is not None:
        tarl[::9] = 0.0
    dino = rng.normal(0, 0.3, (n, ddim)) if ddim else None
    kw = dict(alpha=1.0, theta=0.5 if tdim else 0.0, gamma=0.1 if ddim else 0.0)
    A = api.get_affinity_matrix(pts, tarl, dino, **kw)
    B = ncuts_ref.affinity_sparse(pts, tarl, dino, **kw)
    deg = np.diff(A.indptr)
    assert deg.max() > 400 and (deg > 64).sum() > 500 and (deg < 40).sum() > 2000   # all three regimes are present
    assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
    assert (np.abs(A.data - B.data) / B.data).max() <= 1e-12
    assert abs(A - A.T).max() == 0.0 and np.all(A.diagonal() == 1.0)


@pytest.mark.parametrize("name", ["g2_multicomp_spatial", "g6_connected_tarl"])
def test_lsym_apply_matches_scipy(api, name):
    z, A, _, _ = load(name)
    L, _ = ncuts_ref.laplacian_sym(A)
    x = np.random.default_rng(0).standard_normal(A.shape[0])
    g = api.DeviceGraph.from_scipy(A)
    y = api.lsym_apply(g, x)
    assert np.abs(y - L @ x).max() <= 1e-13


@pytest.mark.parametrize("name", CONNECTED)
def test_fiedler_matches_golden(api, name):
    z, A, _, _ = load(name)
    g = api.DeviceGraph.from_scipy(A)
    lam, ev, iters, resid = api.fiedler(g)
    assert lam == pytest.approx(float(z["eigvals"][1]), rel=1e-8)
    assert resid <= 1e-10 or iters >= A.shape[0] - 1
    assert abs(np.linalg.norm(ev) - 1.0) <= 1e-10
    assert np.abs(np.abs(ev) - z["fiedler_abs"]).max() <= 1e-7
    L, _ = ncuts_ref.laplacian_sym(A)
    assert np.linalg.norm(L @ ev - lam * ev) <= 1e-8
    assert ev[np.argmax(np.abs(ev))] > 0


@pytest.mark.parametrize("name", golden_names())
def test_null_vector_is_in_the_null_space(api, name):
    z, A, _, _ = load(name)
    if int(z["n_components"]) == 1:
        pytest.skip("connected")
    g = api.DeviceGraph.from_scipy(A)
    lam, ev, _, _ = api.fiedler(g)
    L, _ = ncuts_ref.laplacian_sym(A)
    assert lam == 0.0 and abs(np.linalg.norm(ev) - 1.0) <= 1e-12
    assert np.abs(L @ ev).max() <= 1e-13
    ncomp, comp = connected_components(A, directed=False)
    assert np.abs(ev - gpu_model.null_vector(A, ncomp, comp)).max() <= 1e-14


@pytest.mark.parametrize("name", golden_names())
def test_sweep_costs_match_golden(api, name):
    z, A, _, _ = load(name)
    g = api.DeviceGraph.from_scipy(A)
    costs, mask, mcut = api.sweep(g, z["fiedler"])
    ref = z["costs"]
    ok = np.isfinite(ref)
    big = ok & (np.abs(ref) > 1e-9)
    assert np.allclose(costs[big], ref[big], rtol=1e-10, atol=0)
    assert np.abs(costs[ok & ~big] - ref[ok & ~big]).max(initial=0.0) <= 1e-9
    if int(z["n_components"]) == 1:
        assert np.array_equal(mask, z["top_mask"])
        assert mcut == pytest.approx(float(z["top_mcut"]), rel=1e-10)


@pytest.mark.parametrize("name", golden_names())
def test_partition_matches_reference(api, name):
    """normalized_cut on the reference's own CSR: same partition as the imported reference."""
    z, A, _, _ = load(name)
    n = A.shape[0]
    groups = api.normalized_cut(A, n, np.arange(n), T=float(z["T"]), split_lim=0.01)
    lab = ncuts_ref.groups_to_labels(groups, n)
    assert (lab >= 0).all() and sum(len(g) for g in groups) == n
    assert all(np.all(np.diff(g) > 0) for g in groups if len(g) > 1), "members must ascend like labels[mask]"
    assert api.last_stats()["unconverged"] == 0
    assert ncuts_ref.partitions_equal(lab, z["labels"])


@pytest.mark.parametrize("name", golden_names())
def test_groups_equal_model_including_order(api, name):
    z, A, _, _ = load(name)
    n = A.shape[0]
    got = api.normalized_cut(A, n, np.arange(n), T=float(z["T"]))
    exp = gpu_model.normalized_cut_model(A, n, np.arange(n), T=float(z["T"]))
    assert len(got) == len(exp)
    for a, b in zip(got, exp):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("name", CONNECTED)
def test_group_order_matches_reference_on_connected(api, name):
    """Emission order (mask side first) equals the reference's, up to the eigenvector sign."""
    z, A, _, _ = load(name)
    n = A.shape[0]
    got = ncuts_ref.groups_to_labels(api.normalized_cut(A, n, np.arange(n), T=float(z["T"])), n)
    sizes = np.bincount(got)
    assert sorted(sizes.tolist()) == sorted(z["group_sizes"].tolist())


@pytest.mark.parametrize("name", golden_names())
def test_end_to_end_from_points(api, name):
    """Affinity on the device (Morton row order) + recursion, vs the reference labels."""
    z, _, tarl, dino = load(name)
    n = z["points"].shape[0]
    groups = api.ncuts(z["points"], tarl, dino, alpha=float(z["alpha"]), theta=float(z["theta"]), gamma=float(z["gamma"]), T=float(z["T"]))
    lab = ncuts_ref.groups_to_labels(groups, n)
    assert (lab >= 0).all()
    # equal on every fixture, connected or not: a disconnected segment falls into its connected components, which
    # is what the reference's recursion makes of it (none of these fixtures has a <= 1 % remainder of several components)
    assert ncuts_ref.partitions_equal(lab, z["labels"])


def test_edge_cases(api):
    # n <= 2: never split (normalized_cut.py:40 needs W.shape[0] > 2)
    A2 = sp.csr_matrix(np.array([[1.0, 0.5], [0.5, 1.0]]))
    assert [g.tolist() for g in api.normalized_cut(A2, 2, np.arange(2), T=1.0)] == [[0, 1]]
    # split_lim gate on the original count: 5 rows of a 1000-point chunk stay together
    pts = np.array([[0, 0, 0], [0.1, 0, 0], [5, 0, 0], [5.1, 0, 0], [9, 0, 0]], dtype=np.float64)
    A = api.get_affinity_matrix(pts, alpha=1.0, theta=0.0, gamma=0.0)
    assert A.nnz == 9
    assert len(api.normalized_cut(A, 1000, np.arange(5), T=1.0)) == 1
    # isolated points: three singletons + a pair -> every component split off, singletons kept
    groups = api.normalized_cut(A, 5, np.arange(5), T=0.5)
    assert sorted(sorted(g.tolist()) for g in groups) == [[0, 1], [2, 3], [4]]
    # T = 0 never splits (mcut < T is strict, normalized_cut.py:56)
    assert len(api.normalized_cut(A, 5, np.arange(5), T=0.0)) == 1
    # errors mirror the reference: gamma without DINO features
    with pytest.raises(ValueError):
        api.get_affinity_matrix(pts, alpha=1.0, theta=0.0, gamma=0.1)
    # labels are returned as given
    out = api.normalized_cut(A, 5, np.array([10, 11, 12, 13, 14]), T=0.5)
    assert sorted(x for g in out for x in g.tolist()) == [10, 11, 12, 13, 14]


def test_run_to_run_reproducible(api):
    from autoinst_amd import synth
    ch = synth.synthetic_chunk(20_000, seed=5)
    a = api.ncuts(ch["points"], ch["tarl"], alpha=1.0, theta=0.5, gamma=0.0, T=0.03)
    b = api.ncuts(ch["points"], ch["tarl"], alpha=1.0, theta=0.5, gamma=0.0, T=0.03)
    assert len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))


def test_repeated_calls_freeze_every_segment_at_the_same_size_of_T(api):
    """The sizes of T at which a segment is judged are a function of its own history, not of timing: over repeated calls the
    labels AND the work counter (sum over solves of rows x Lanczos steps) are identical.  (Round 3 found the convergence check
    reading its pool entry through the scalar cache: once in ~50 000 solves the waves of a block disagreed on the segment and
    the solve froze at another size of T; `tools/soak_trace.py` is the long version of this test.)"""
    from autoinst_amd import synth
    chunks = [synth.synthetic_chunk(n, seed, tarl=True) for n, seed in ((30000, 1), (24000, 2), (30000, 3), (18000, 4))]
    first = None
    for r in range(25):
        graphs = [api.build_affinity(c["points"], c["tarl"], alpha=1.0, theta=0.5, gamma=0.0) for c in chunks]
        labs, ngs, st = api.ncuts_labels_batch(graphs, None, 0.03)
        for g in graphs:
            g.free()
        cur = (labs, ngs, st["spmv_rows"], st["spmv_nnz"], st["lanczos_solves"], st["max_resid"])
        if first is None:
            first = cur
            continue
        assert cur[2:] == first[2:], (r, cur[2:], first[2:])
        assert cur[1] == first[1] and all(np.array_equal(a, b) for a, b in zip(cur[0], first[0]))


@pytest.mark.gpu
def test_workspace_settles_in_one_block(api):
    """ai_ctx_mem_info: calls of growing size append workspace blocks; the next call starts by replacing the list with one
    block of the largest need seen, and the results do not change."""
    from autoinst_amd import synth
    ctx = api.Context()
    chunks = [synth.synthetic_chunk(n, seed, tarl=True) for n, seed in ((5000, 1), (40000, 2), (12000, 3))]
    labs = []
    for c in chunks + chunks[:1]:
        g = api.build_affinity(c["points"], c["tarl"], alpha=1.0, theta=0.5, gamma=0.0, ctx=ctx)
        labs.append(api.ncuts_labels(g, g.n, 0.03)[0])
        g.free()
    info = ctx.mem_info()
    assert info["workspace_blocks"] == 1 and info["workspace"] > 0 and info["graphs_live"] == 0 and info["graphs_kept"] > 0
    assert np.array_equal(labs[0], labs[3])
    ref = api.ncuts_labels(api.build_affinity(chunks[1]["points"], chunks[1]["tarl"], alpha=1.0, theta=0.5, gamma=0.0), 40000, 0.03)[0]
    assert np.array_equal(labs[1], ref)
    ctx.close()


@pytest.mark.gpu
def test_batched_chunks_equal_separate_calls(api):
    """ai_ncut_batch: several chunks as root segments of one frontier give each chunk's own result."""
    from autoinst_amd import synth
    chunks = [synth.synthetic_chunk(n, seed, tarl=True) for n, seed in ((9000, 1), (4000, 2), (15000, 3))]
    graphs = [api.build_affinity(c["points"], c["tarl"], alpha=1.0, theta=0.5, gamma=0.0) for c in chunks]
    single = [api.ncuts_labels(g, g.n, 0.03) for g in graphs]
    labs, ngs, st = api.ncuts_labels_batch(graphs, None, 0.03)
    assert st["unconverged"] == 0
    for (l1, n1, _), l2, n2 in zip(single, labs, ngs):
        assert n1 == n2 and np.array_equal(l1, l2)
    # a chunk below the split limit of ITS OWN original size stays whole, the others are unaffected
    labs2, ngs2, _ = api.ncuts_labels_batch(graphs, [g.n for g in graphs[:2]] + [10 ** 9], 0.03)
    assert ngs2[2] == 1 and np.all(labs2[2] == 0) and np.array_equal(labs2[0], labs[0])


@pytest.mark.gpu
def test_admission_window_and_many_chunks_in_one_call(api):
    """The asynchronous frontier admits the chunks of a call as far as `window_rows` allows and the rest as earlier chunks
    finish; every chunk's labels are those of its own call, whatever the window (the sizes of T at which a segment may
    freeze depend on its own step count only).  40 chunks of the real size range in ONE call (a whole map)."""
    from autoinst_amd import synth
    rng = np.random.default_rng(7)
    sizes = [int(x) for x in np.exp(rng.uniform(np.log(3000), np.log(12000), 40))]
    chunks = [synth.synthetic_chunk(n, 100 + i, tarl=True) for i, n in enumerate(sizes)]
    graphs = [api.build_affinity(c["points"], c["tarl"], alpha=1.0, theta=0.5, gamma=0.0) for c in chunks]
    single = [api.ncuts_labels(g, g.n, 0.03) for g in graphs[:6]]
    labs_all, ngs_all, st_all = api.ncuts_labels_batch(graphs, None, 0.03)
    labs_win, ngs_win, st_win = api.ncuts_labels_batch(graphs, None, 0.03, window_rows=30_000)   # 3 - 8 chunks at a time
    labs_one, ngs_one, _ = api.ncuts_labels_batch(graphs, None, 0.03, window_rows=1)             # one chunk at a time
    for g in graphs:
        g.free()
    assert st_all["unconverged"] == 0 and st_win["unconverged"] == 0
    assert ngs_all == ngs_win == ngs_one
    for a, b, c in zip(labs_all, labs_win, labs_one):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    for (l1, n1, _), l2, n2 in zip(single, labs_all, ngs_all):
        assert n1 == n2 and np.array_equal(l1, l2)
    assert st_win["lanczos_steps"] > st_all["lanczos_steps"]   # the window really serialised the chunks


def _run_hook_case(case):
    """The fault-injection hooks (AI_FLOW_INJECT, AI_FLOW_INJECT_HIST, AI_FLOW_TRUE_LIMIT, AI_FLOW_SMAX) exist only in the test-only build
    (libautoinst_hip_lockstep.so: -DAI_WITH_LOCKSTEP -DAI_TEST_HOOKS; the shipped library reads nothing from the environment that can
    change a result), so the cases that need them run in a child process that loads that build: tests/hook_cases.py."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    locklib = os.path.join(root, "autoinst_amd", "libautoinst_hip_lockstep.so")
    if not os.path.exists(locklib):
        pytest.skip("libautoinst_hip_lockstep.so is not built (make -C autoinst_amd/csrc lockstep)")
    env = dict(os.environ, AUTOINST_HIP_LIB=locklib)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "hook_cases.py"), case], env=env, timeout=600, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert f"hook case {case}: ok" in r.stdout, r.stdout[-2000:]


@pytest.mark.gpu
def test_hundreds_of_chunks_in_one_call_and_children_that_wait_for_a_slot(api):
    """Admission is slot-aware (round-3 advisor finding): a call whose live segments outnumber the Lanczos slots / pool records it
    was set up with lets children wait for a slot instead of failing with an internal error.  (i) 600 small chunks in ONE call;
    (ii) chunks whose num_points_orig is far below their size, so that the 1 % rule allows thousands of live segments -- both with the
    shipped library; (iii) the same labels when a call has only 48 / 32 slots, so that every wave defers children (AI_FLOW_SMAX, a hook
    of the test-only build: tests/hook_cases.py `slots`)."""
    from autoinst_amd import synth
    rng = np.random.default_rng(11)
    sizes = [int(x) for x in np.exp(rng.uniform(np.log(1500), np.log(5000), 600))]
    chunks = [synth.synthetic_chunk(n, 500 + i, tarl=False) for i, n in enumerate(sizes)]
    graphs = [api.build_affinity(c["points"], None, alpha=1.0, theta=0.0, gamma=0.0) for c in chunks]
    labs, ngs, st = api.ncuts_labels_batch(graphs, None, 0.075)
    assert st["unconverged"] == 0 and len(labs) == 600
    for i in (0, 17, 311, 599):
        l1, n1, _ = api.ncuts_labels(graphs[i], graphs[i].n, 0.075)
        assert n1 == ngs[i] and np.array_equal(l1, labs[i])
    big = graphs[:8]
    norig = [max(1, g.n // 100) for g in big]
    labs_s, ngs_s, st_s = api.ncuts_labels_batch(big, norig, 0.075)
    assert st_s["unconverged"] == 0
    for g, lab, ng in zip(big, labs_s, ngs_s):
        assert lab.shape == (g.n,) and lab.min() == 0 and lab.max() == ng - 1 and len(np.unique(lab)) == ng
    for g in graphs:
        g.free()
    _run_hook_case("slots")


@pytest.mark.gpu
def test_a_spoiled_ritz_pair_is_caught_by_the_true_residual_and_solved_again(api):
    """Every harvested Ritz pair is tested against the segment's own operator (fk_resid: ||M v - theta v|| <= 1e-6 ||v||) before it
    is cut; the packed histories carry the device's size of T and a checksum.  A healthy call of the shipped library restarts nothing
    and reports its largest true residual; the three fences themselves are exercised with the hooks of the test-only build
    (tests/hook_cases.py `fences`)."""
    from autoinst_amd import synth
    ch = synth.synthetic_chunk(30000, 21, tarl=True)
    g = api.build_affinity(ch["points"], ch["tarl"], alpha=1.0, theta=0.5, gamma=0.0)
    lab0, ng0, st0 = api.ncuts_labels(g, g.n, 0.03)
    assert st0["restarted_solves"] == 0 and st0["unconverged"] == 0 and st0["hist_retries"] == 0
    assert st0["accepted_above_limit"] == 0 and st0["check_timeouts"] == 0
    assert st0["true_resid_limit"] == 1e-6
    # the estimate that stopped the solves and the truth agree (both <= tol = 1e-10 on this chunk, up to rounding in the residual itself)
    assert 0.0 < st0["max_true_resid"] <= 2e-10, st0
    g.free()
    _run_hook_case("fences")


@pytest.mark.gpu
def test_warm_start_and_hash_start_give_the_same_labels(api, monkeypatch):
    """Since round 5 a segment with a solved ancestor starts its Lanczos solve from that ancestor's second Ritz vector (carried to its
    rows by the partitions) instead of the hash vector alone; AI_FLOW_WARM=0 is the hash start of rounds 1-4.  Another start vector,
    the same eigenvector: the labels are equal -- single chunks (with disconnected segments, whose components inherit through the
    split), a batched call, a tri-modal chunk -- and the warm start takes fewer steps."""
    from autoinst_amd import synth
    cases = [(20000, 3, "tarl", 0.03), (50000, 0, "spatial", 0.075), (30000, 21, "tarl", 0.03), (20000, 2, "tri", 0.005)]
    graphs = []
    for n, seed, mode, T in cases:
        ch = synth.synthetic_chunk(n, seed, tarl=mode != "spatial", dino=mode == "tri")
        graphs.append((api.build_affinity(ch["points"], ch["tarl"] if mode != "spatial" else None, ch["dino"] if mode == "tri" else None,
                                          alpha=1.0, theta=0.0 if mode == "spatial" else 0.5, gamma=0.1 if mode == "tri" else 0.0), T))
    warm = [api.ncuts_labels(g, g.n, T) for g, T in graphs]
    warm_b = api.ncuts_labels_batch([g for g, _ in graphs[:3:2]], None, 0.03)
    monkeypatch.setenv("AI_FLOW_WARM", "0")
    steps_w = steps_h = 0
    for (g, T), (lab1, ng1, st1) in zip(graphs, warm):
        lab0, ng0, st0 = api.ncuts_labels(g, g.n, T)
        assert ng0 == ng1 and np.array_equal(lab0, lab1), (g.n, ng0, ng1)
        assert st0["lanczos_solves"] == st1["lanczos_solves"] and st0["unconverged"] == st1["unconverged"] == 0
        assert st1["max_true_resid"] <= 2e-10 and st0["max_true_resid"] <= 2e-10
        steps_w += st1["spmv_rows"]
        steps_h += st0["spmv_rows"]
    hash_b = api.ncuts_labels_batch([g for g, _ in graphs[:3:2]], None, 0.03)
    monkeypatch.delenv("AI_FLOW_WARM")
    assert hash_b[1] == warm_b[1] and all(np.array_equal(a, b) for a, b in zip(hash_b[0], warm_b[0]))
    assert steps_w < 0.97 * steps_h, (steps_w, steps_h)   # rows x steps: ~8 % fewer
    for g, _ in graphs:
        g.free()


@pytest.mark.gpu
def test_level_synchronous_driver_gives_the_same_labels(tmp_path):
    """The level-synchronous driver of rounds 1-2 (test-only build libautoinst_hip_lockstep.so, AI_NCUT_LOCKSTEP=1) and the
    asynchronous frontier of the shipped library give identical labels: same solver arithmetic per segment, same sweep, same
    emission order.  The shipped library refuses AI_NCUT_LOCKSTEP=1."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    locklib = os.path.join(root, "autoinst_amd", "libautoinst_hip_lockstep.so")
    if not os.path.exists(locklib):
        pytest.skip("libautoinst_hip_lockstep.so is not built (make -C autoinst_amd/csrc lockstep)")
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from autoinst_amd import ncuts_api as api, synth\n"
        "out = []\n"
        "for n, seed in ((20000, 3), (6000, 4)):\n"
        "    ch = synth.synthetic_chunk(n, seed, tarl=True)\n"
        "    g = api.build_affinity(ch['points'], ch['tarl'], alpha=1.0, theta=0.5, gamma=0.0)\n"
        "    lab, ng, st = api.ncuts_labels(g, n, 0.03)\n"
        "    out.append(lab)\n"
        "np.savez(sys.argv[1], *out)\n"
    ) % root
    res = {}
    for mode in ("0", "1"):
        path = str(tmp_path / f"labels_{mode}.npz")
        env = dict(os.environ, AI_NCUT_LOCKSTEP=mode)
        if mode == "1":
            env["AUTOINST_HIP_LIB"] = locklib
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=300)
        res[mode] = np.load(path)
    for k in res["0"].files:
        assert np.array_equal(res["0"][k], res["1"][k])



def test_radius_boundary_and_duplicates(api):
    """`spatial_distance <= PROXIMITY_THRESHOLD` is inclusive (ncuts_utils.py:61); duplicates have weight 1."""
    eps = np.nextafter(1.0, 2.0)
    pts = np.array([[0, 0, 0], [1.0, 0, 0], [0, eps, 0], [0, 0, 0], [0.6, 0.8, 0.0]], dtype=np.float64)
    A = api.get_affinity_matrix(pts, alpha=1.0, theta=0.0, gamma=0.0).toarray()
    ref = ncuts_ref.affinity_dense(pts, alpha=1.0, theta=0.0, gamma=0.0)
    assert np.array_equal(A != 0, ref != 0)
    assert A[0, 1] == np.exp(-1.0) and A[0, 2] == 0.0 and A[0, 3] == 1.0 and A[0, 4] != 0.0
    assert np.abs(A - ref).max() <= 1e-15
    # alpha = 0 drops the spatial factor (ncuts_utils.py:63-66): pure 0/1 mask
    M = api.get_affinity_matrix(pts, alpha=0.0, theta=0.0, gamma=0.0).toarray()
    assert np.array_equal(M, (ref != 0).astype(float))


def test_all_zero_tarl_rows_have_no_tarl_penalty(api):
    from autoinst_amd import synth
    pts, gt = synth.surface_chunk(1500, seed=9, extent=9.0)
    f = synth.surrogate_features(gt, 96, 9, zero_frac=0.5)
    A = api.get_affinity_matrix(pts, f, alpha=1.0, theta=0.5, gamma=0.0)
    S = api.get_affinity_matrix(pts, alpha=1.0, theta=0.0, gamma=0.0)
    z = ~f.any(1)
    assert z.sum() > 100
    Ad, Sd = A.toarray(), S.toarray()
    assert np.array_equal(Ad[z], Sd[z]) and np.array_equal(Ad[:, z], Sd[:, z])   # t = 0 wherever a row has no feature
    assert (Ad[~z][:, ~z] < Sd[~z][:, ~z]).any()
    assert np.abs(Ad - ncuts_ref.affinity_dense(pts, f, alpha=1.0, theta=0.5, gamma=0.0)).max() <= 1e-14


def test_20k_device_equals_model_exactly(api):
    """Same algorithm, two implementations: HIP path == NumPy model on a 20k chunk (groups and order)."""
    from autoinst_amd import synth
    ch = synth.synthetic_chunk(20_000, seed=0, tarl=True)
    A = ncuts_ref.affinity_sparse(ch["points"], ch["tarl"], alpha=1.0, theta=0.5, gamma=0.0)
    n = A.shape[0]
    got = api.normalized_cut(A, n, np.arange(n), T=0.03)
    exp = gpu_model.normalized_cut_model(A, n, np.arange(n), T=0.03)
    assert len(got) == len(exp) and all(np.array_equal(a, b) for a, b in zip(got, exp))


def _dense_blobs(n, spread, seed):
    """Two Gaussian blobs whose 1 m radius graph has ~n / 8 .. n / 2 neighbours per point."""
    rng = np.random.default_rng(seed)
    a = rng.normal(0.0, spread, (n // 2, 3))
    b = rng.normal(0.0, spread, (n - n // 2, 3)) + np.array([2.2 * spread + 0.6, 0.0, 0.0])
    return np.concatenate([a, b])


@pytest.mark.parametrize("n,spread", [(2500, 0.9), (3000, 0.45), (6000, 2.0)])
def test_staged_gather_equals_plain_gather(api, monkeypatch, n, spread):
    """The SpMV with LDS-staged distinct columns (entries taken as aligned quads) and the plain global-memory gather
    (entries strided over the lanes) add a row's products in different orders: identical labels, residuals equal to
    rounding.  The dense cases exceed the encoder's
    per-task capacity (4096 entries / 1024 distinct columns), so their tasks take the plain path inside
    the staged kernel; the sparse case is fully encoded."""
    pts = _dense_blobs(n, spread, n)
    g = api.build_affinity(pts, None, alpha=1.0, theta=0.0, gamma=0.0)
    assert g.nnz / g.n > (30 if spread > 1.5 else 128)
    monkeypatch.setenv("AI_SPMV_VARIANT", "0")
    l0, n0, s0 = api.ncuts_labels(g, n, 0.5)
    monkeypatch.setenv("AI_SPMV_VARIANT", "1")
    l1, n1, s1 = api.ncuts_labels(g, n, 0.5)
    A = g.to_scipy()
    g.free()
    assert n0 == n1 >= 2 and np.array_equal(l0, l1)
    # (lanczos_steps counts launches, including the few issued past convergence while a check was in flight)
    assert s0["lanczos_solves"] == s1["lanczos_solves"] > 0 and s0["unconverged"] == s1["unconverged"] == 0
    assert abs(s0["max_resid"] - s1["max_resid"]) <= 0.5 * s1["max_resid"] and max(s0["max_resid"], s1["max_resid"]) <= 1e-9   # the two gather forms add in different orders: ~1e-16 in T, visible in a 1e-10 residual
    if spread < 1.5:   # (T = 0.5 on the sparse blob pair recurses into near-tie cuts: no model comparison there)
        exp = gpu_model.normalized_cut_model(A, n, np.arange(n), T=0.5)
        lab = np.empty(n, np.int64)
        for k, grp in enumerate(exp):
            lab[grp] = k
        assert ncuts_ref.partitions_equal(l0, lab)


def test_concurrent_contexts_give_the_sequential_results(api):
    """bench.py's arrangement: several host threads, one Context each, batched calls in flight together
    (the C calls release the GIL) -- every chunk's labels equal those of a plain sequential call."""
    import threading
    from autoinst_amd import synth
    chunks = [synth.synthetic_chunk(n, seed, tarl=True) for n, seed in ((7000, 21), (12000, 22), (5000, 23), (9000, 24), (16000, 25), (6000, 26))]
    ref = []
    for c in chunks:
        g = api.build_affinity(c["points"], c["tarl"], alpha=1.0, theta=0.5, gamma=0.0)
        ref.append(api.ncuts_labels(g, g.n, 0.03)[0])
        g.free()
    out = [None] * 3
    err = []

    def work(k):
        try:
            ctx = api.Context(0)
            for _ in range(3):   # repeated, so that the threads really overlap and buffers get re-used
                graphs = [api.build_affinity(c["points"], c["tarl"], alpha=1.0, theta=0.5, gamma=0.0, ctx=ctx) for c in chunks[2 * k:2 * k + 2]]
                labs, _, st = api.ncuts_labels_batch(graphs, None, 0.03)
                for g in graphs:
                    g.free()
                assert st["unconverged"] == 0
            out[k] = labs
            ctx.close()
        except BaseException as e:   # noqa: BLE001 - reported in the main thread
            err.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    for k in range(3):
        for b in range(2):
            assert np.array_equal(out[k][b], ref[2 * k + b])


@pytest.mark.parametrize("beta,gamma", [(0.0, 0.0), (0.7, 0.0), (0.0, 0.1), (0.7, 0.1)])
def test_ncuts_chunk_drop_in_with_stand_in_pipeline_modules(api, monkeypatch, beta, gamma):
    """`ncuts_chunk` has the reference's signature and 5-tuple (ncuts_utils.py:28-204).  The reference's
    surroundings (open3d, its `config` and `utils` packages) are not importable here, so minimal
    stand-ins with the same names are injected for this test only: what is checked is the glue around
    the hot path -- arguments consumed, groups painted, colours re-projected onto the fine cloud."""
    import sys
    import types
    from autoinst_amd import synth

    class PC:
        def __init__(self, points, colors=None):
            self.points = np.asarray(points, dtype=np.float64)
            self.colors = np.zeros_like(self.points) if colors is None else np.asarray(colors, dtype=np.float64)

        def paint_uniform_color(self, c):
            self.colors = np.tile(np.asarray(c, dtype=np.float64), (self.points.shape[0], 1))

        def __add__(self, o):
            return PC(np.concatenate([self.points, o.points]), np.concatenate([self.colors, o.colors]))

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        monkeypatch.setitem(sys.modules, name, m)
        return m

    ch = synth.synthetic_chunk(6000, 31, tarl=True, dino=True)
    rng = np.random.default_rng(0)
    sam_ids = rng.integers(0, 4, (6000, 5))
    sam_ids[rng.random((6000, 5)) < 0.3] = -1

    def image_features(dataset, pcd, chunk_indices, chunk_nc, T_pcd, cam_idx, sam=False, dino=False, pcd_chunk=None):
        # the three return shapes of utils/image/image_utils.py's image_based_features_per_patch
        calls["image"] = (sam, dino, pcd_chunk is not None)
        if sam and not dino:
            return [sam_ids]
        if dino and not sam:
            return [ch["dino"]], None
        return [sam_ids], [ch["dino"]]

    fine = np.concatenate([ch["points"] + rng.normal(0, 0.03, ch["points"].shape) for _ in range(2)])
    ground = np.stack([rng.uniform(-10, 10, 400), rng.uniform(-10, 10, 400), rng.normal(-1.5, 0.02, 400)], 1)
    calls = {}
    mod("open3d", utility=types.SimpleNamespace(Vector3dVector=lambda a: np.asarray(a)))
    mod("config", CONFIG=dict(alpha=1.0, beta=beta, gamma=gamma, theta=0.5, T=0.03), PROXIMITY_THRESHOLD=1.0, SPLIT_LIM=0.01,
        ADJACENT_FRAMES_CAM=(16, 13), ADJACENT_FRAMES_TARL=(10, 10), MEAN_HEIGHT=0.6, CHUNK_SIZE=np.array([25, 25, 25]),
        MAJOR_VOXEL_SIZE=0.35, TARL_NORM=False)
    mod("utils")
    mod("utils.image")
    mod("utils.image.image_utils", dinov2_mean=lambda p2d: p2d, image_based_features_per_patch=image_features)
    mod("utils.point_cloud")
    mod("utils.point_cloud.chunk_generation",
        get_indices_feature_reprojection=lambda idx, first, adjacent_frames: (list(idx[:3]), None))
    mod("utils.point_cloud.point_cloud_utils",
        get_statistical_inlier_indices=lambda pcd: np.arange(pcd.points.shape[0]),
        get_subpcd=lambda pcd, idx: PC(pcd.points[idx], pcd.colors[idx]),
        transform_pcd=lambda pts, T: np.asarray(pts) @ T[:3, :3].T + T[:3, 3])

    class Dataset:
        """Three lidar scans whose points carry the surrogate TARL feature of the nearest chunk point
        (float32, as the .bin files hold them), each in its own sensor frame."""
        def __init__(self):
            r = np.random.default_rng(5)
            self.scans = {}
            for k in range(3):
                sel = r.choice(ch["points"].shape[0], 9000, replace=True)
                world = ch["points"][sel] + r.normal(0, 0.05, (9000, 3))
                pose = np.eye(4)
                pose[:3, 3] = [0.5 * k, -0.3 * k, 0.1]
                self.scans[k] = (world - pose[:3, 3], ch["tarl"][sel].astype(np.float32), pose)

        def get_tarl_features(self, i):
            calls["tarl"] = calls.get("tarl", 0) + 1
            return self.scans[i][1]

        def get_point_cloud(self, i):
            return self.scans[i][0]

        def get_pose(self, i):
            return self.scans[i][2]

    dataset = Dataset()
    from oracle import points_ref
    world = [dataset.scans[k][0] + dataset.scans[k][2][:3, 3] for k in range(3)]
    inbox = [np.all(w > -12.5, axis=1) & np.all(w < 12.5, axis=1) for w in world]
    pooled = points_ref.tarl_pool(ch["points"], np.concatenate([w[m] for w, m in zip(world, inbox)]),
                                  np.concatenate([dataset.scans[k][1][m] for k, m in enumerate(inbox)]), radius=0.175)
    mod("utils.visualization_utils", generate_random_colors=lambda n: [(int(37 * i) % 256, int(91 * i) % 256, 1 + i % 255) for i in range(n)])

    chunk_major, pcd_chunk, pcd_ground = PC(ch["points"]), PC(fine), PC(ground)
    d = {"center_ids": [5], "center_positions": [np.zeros(3)], "indices": [np.arange(fine.shape[0])],
         "pcd_nonground_chunks": [pcd_chunk], "pcd_ground_chunks": [pcd_ground],
         "pcd_nonground_chunks_major_downsampling": [chunk_major],
         "kitti_labels": {"ground": {"instance": [np.arange(400)], "semantic": [np.full(400, 40)]}}}
    merged, chunk_out, cut, inst_g, seg_g = api.ncuts_chunk(dataset, d, None, np.eye(4), list(range(20)), sequence=0, patchwise_indices=[[3, 4]])
    assert calls.get("tarl") == 3 and chunk_out is pcd_chunk
    assert calls.get("image") == ((bool(beta), bool(gamma), bool(gamma) and not beta) if (beta or gamma) else None)
    # the fine cloud carries the colour of the nearest major-voxel point; groups are painted with distinct colours
    groups = api.ncuts(ch["points"], pooled, ch["dino"] if gamma else None, sam=sam_ids if beta else None, alpha=1.0, theta=0.5,
                       gamma=gamma, beta=beta, T=0.03)
    assert np.unique(chunk_out.colors, axis=0).shape[0] == len(groups)
    from scipy.spatial import cKDTree
    nn = cKDTree(ch["points"]).query(fine)[1]
    lab = np.empty(ch["points"].shape[0], np.int64)
    for k, g in enumerate(groups):
        lab[g] = k
    col_of = {k: chunk_out.colors[np.flatnonzero(lab[nn] == k)[0]] for k in range(len(groups))}
    assert all(np.array_equal(chunk_out.colors[i], col_of[lab[nn[i]]]) for i in range(0, fine.shape[0], 97))
    assert merged.points.shape[0] == fine.shape[0] + cut.points.shape[0] and np.all(cut.colors == 0)
    assert inst_g.shape == seg_g.shape == (cut.points.shape[0],)


def test_c_abi_rejects_bad_arguments(api):
    """Status -1 (bad argument) surfaces as ValueError with the library's message, never as a crash."""
    import ctypes as C
    from autoinst_amd import _ffi, labels_api, points_api
    lib = _ffi.load()
    ctx = api.default_context()
    pts = np.zeros((4, 3))
    with pytest.raises(ValueError, match="radius"):
        api.get_affinity_matrix(pts, alpha=1.0, theta=0.0, gamma=0.0, radius=0.0)
    with pytest.raises(ValueError):
        api.get_affinity_matrix(pts, np.zeros((4, 96)), alpha=1.0, theta=0.5, gamma=0.1)      # gamma without DINO
    h = C.c_void_p()
    assert lib.ai_affinity_build(ctx._h, None, 4, None, 0, None, 0, 1.0, 0.0, 0.0, 1.0, 0, C.byref(h)) == -1
    assert b"ai_affinity_build" in lib.ai_last_error()
    g = api.build_affinity(np.random.default_rng(0).normal(0, 1, (50, 3)), None, alpha=1.0, theta=0.0, gamma=0.0)
    ng = C.c_int32()
    assert lib.ai_ncut(ctx._h, g._h, 50, 0.1, 0.01, None, None, C.byref(ng), None) == -1          # labels_out missing
    with pytest.raises(ValueError):
        api.eigs_smallest(g, 0)
    with pytest.raises(ValueError):
        api.eigs_smallest(g, 65)
    g.free()
    with pytest.raises(ValueError):
        labels_api.label_pairs(np.zeros(3, np.int32), np.zeros(4, np.int32))
    with pytest.raises(ValueError):
        labels_api.merge_associate(np.zeros((3, 3)), np.zeros(3, np.int32), np.zeros((2, 3)), np.zeros(2, np.int32), np.zeros(3), 0, 2)
    with pytest.raises(ValueError):
        points_api.tarl_pool(np.zeros((3, 3)), np.zeros((5, 3)), np.zeros((4, 96), np.float32))


def test_sam_factor_matches_oracle(api):
    """Row a5: exp(-beta * fraction of co-labelled views with differing SAM ids), multiplied between the
    spatial and the DINO factor (ncuts_utils.py:151-156)."""
    rng = np.random.default_rng(12)
    n = 4000
    pts = rng.normal(0, 3.0, (n, 3))
    tarl = rng.normal(0, 1, (n, 96))
    tarl[::13] = 0.0
    dino = rng.normal(0, 1, (n, 384))
    sam = rng.integers(0, 5, (n, 6))
    sam[rng.random((n, 6)) < 0.35] = -1
    for kw in (dict(alpha=1.0, theta=0.5, gamma=0.0, beta=0.7), dict(alpha=1.0, theta=0.5, gamma=0.1, beta=1.3),
               dict(alpha=0.0, theta=0.0, gamma=0.0, beta=2.0)):
        A = api.get_affinity_matrix(pts, tarl, dino, sam=sam, **kw)
        B = ncuts_ref.affinity_sparse(pts, tarl, dino, sam=sam, **kw)
        assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
        assert (np.abs(A.data - B.data) / B.data).max() <= 1e-12
        assert abs(A - A.T).max() == 0.0 and np.all(A.diagonal() == 1.0)
    with pytest.raises(ValueError):
        api.get_affinity_matrix(pts, tarl, None, alpha=1.0, theta=0.5, gamma=0.0, beta=0.5)
    groups = api.ncuts(pts, tarl, None, sam=sam, alpha=1.0, theta=0.5, gamma=0.0, beta=0.7, T=0.03)
    assert sorted(np.concatenate(groups).tolist()) == list(range(n))


def test_random_small_clouds_device_equals_model(api):
    """Forty random clouds (50-3000 points, blobs / sheets / lines, several T and radii): the HIP path
    and the NumPy model of the same algorithm give the same partition (small segments take the
    dense-check path, sparse rows exhaust the encoder's pool, single points stay single)."""
    rng = np.random.default_rng(2024)
    bad = []
    for case in range(40):
        n = int(rng.integers(50, 3000))
        kind = case % 4
        if kind == 0:      # a few blobs
            c = rng.normal(0, 4, (int(rng.integers(2, 7)), 3))
            pts = c[rng.integers(0, c.shape[0], n)] + rng.normal(0, rng.uniform(0.3, 1.2), (n, 3))
        elif kind == 1:    # a bumpy sheet
            xy = rng.uniform(-8, 8, (n, 2))
            pts = np.c_[xy, 0.3 * np.sin(xy[:, 0]) + rng.normal(0, 0.05, n)]
        elif kind == 2:    # line segments (sparse rows)
            t = rng.uniform(0, 30, n)
            pts = np.c_[t, np.floor(t / 10) * 3.0 + rng.normal(0, 0.05, n), rng.normal(0, 0.05, n)]
        else:              # uniform box with isolated points
            pts = rng.uniform(-6, 6, (n, 3))
        T = float(rng.choice([0.01, 0.05, 0.2]))
        A = ncuts_ref.affinity_sparse(pts, None, alpha=1.0, theta=0.0, gamma=0.0)
        got = api.normalized_cut(A, n, np.arange(n), T=T)
        exp = gpu_model.normalized_cut_model(A, n, np.arange(n), T=T)
        la, lb = ncuts_ref.groups_to_labels(got, n), ncuts_ref.groups_to_labels(exp, n)
        if not ncuts_ref.partitions_equal(la, lb):
            bad.append((case, n, kind, T, len(got), len(exp), ncuts_ref.adjusted_rand_index(la, lb)))
    assert not bad, bad


def test_pure_c_caller_gets_the_same_labels(api, tmp_path):
    """The boundary is the C ABI: a C program (no Python, no torch in the process) builds the graph and runs
    the recursion through include/autoinst_hip.h and gets the labels the Python mirror returns."""
    import shutil
    import subprocess
    from autoinst_amd import _ffi, synth
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    ch = synth.synthetic_chunk(8000, 41, tarl=True)
    n = ch["points"].shape[0]
    (tmp_path / "pts.bin").write_bytes(np.ascontiguousarray(ch["points"]).tobytes())
    (tmp_path / "tarl.bin").write_bytes(np.ascontiguousarray(ch["tarl"]).tobytes())
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include <stdio.h>
#include <stdlib.h>
#include "autoinst_hip.h"
static void* slurp(const char* path, size_t bytes) {
  void* p = malloc(bytes); FILE* f = fopen(path, "rb");
  if (!p || !f || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
  fclose(f); return p;
}
int main(int argc, char** argv) {
  long n = atol(argv[1]);
  double* pts = slurp(argv[2], (size_t)n * 3 * sizeof(double));
  double* tarl = slurp(argv[3], (size_t)n * 96 * sizeof(double));
  ai_ctx* ctx = 0; ai_csr* g = 0;
  if (ai_ctx_create(0, &ctx)) { fprintf(stderr, "%s\n", ai_last_error()); return 1; }
  if (ai_affinity_build(ctx, pts, n, tarl, 96, 0, 0, 1.0, 0.5, 0.0, 1.0, AI_MEM_HOST, &g)) { fprintf(stderr, "%s\n", ai_last_error()); return 1; }
  int32_t* lab = malloc((size_t)n * sizeof(int32_t)); int32_t ng = 0; ai_ncut_stats st;
  if (ai_ncut(ctx, g, n, 0.03, 0.01, 0, lab, &ng, &st)) { fprintf(stderr, "%s\n", ai_last_error()); return 1; }
  FILE* f = fopen(argv[4], "wb"); fwrite(lab, sizeof(int32_t), (size_t)n, f); fclose(f);
  printf("%d %lld\n", ng, (long long)st.unconverged);
  ai_csr_free(ctx, g); ai_ctx_destroy(ctx);
  return 0;
}
''')
    libdir = os.path.dirname(_ffi.LIB_PATH)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "caller"
    r = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", str(exe), "-L", libdir, "-lautoinst_hip",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe), str(n), str(tmp_path / "pts.bin"), str(tmp_path / "tarl.bin"), str(tmp_path / "lab.bin")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    ng_c, unconv = (int(x) for x in out.stdout.split())
    lab_c = np.fromfile(tmp_path / "lab.bin", dtype=np.int32)
    g = api.build_affinity(ch["points"], ch["tarl"], alpha=1.0, theta=0.5, gamma=0.0)
    lab_py, ng_py, _ = api.ncuts_labels(g, n, 0.03)
    g.free()
    assert unconv == 0 and ng_c == ng_py and np.array_equal(lab_c, lab_py)


def test_two_cameras_match_oracle(api):
    """Lists of per-camera matrices for the DINO and SAM factors (the reference loops over cameras; its config has
    one): the extra cameras multiply the finished graph, values agree with the oracle to 1e-12."""
    rng = np.random.default_rng(21)
    n = 3000
    pts = rng.normal(0, 2.5, (n, 3))
    tarl = rng.normal(0, 1, (n, 96))
    dino = [rng.normal(0, 1, (n, 384)), rng.normal(0, 1, (n, 384)), rng.normal(0, 1, (n, 384))]
    sam = [rng.integers(-1, 4, (n, 5)), rng.integers(-1, 3, (n, 2))]
    kw = dict(alpha=1.0, theta=0.5, gamma=0.05, beta=0.8)
    A = api.get_affinity_matrix(pts, tarl, dino, sam=sam, **kw)
    B = ncuts_ref.affinity_sparse(pts, tarl, dino, sam=sam, **kw)
    assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
    assert (np.abs(A.data - B.data) / B.data).max() <= 1e-12
    assert abs(A - A.T).max() == 0.0 and np.all(A.diagonal() == 1.0)
    one = api.get_affinity_matrix(pts, tarl, dino[:1], sam=sam[:1], **kw)
    assert (one.data >= A.data).all() and (one.data > A.data).any()


def test_no_convergence_is_an_error_like_the_reference(api):
    """A step cap too small for the residual tolerance: status -3 -> NoConvergence (a RuntimeError, as scipy's
    ArpackNoConvergence is at normalized_cut.py:49)."""
    from autoinst_amd import _ffi, synth
    ch = synth.synthetic_chunk(6000, 3, tarl=False)
    g = api.build_affinity(ch["points"], None, alpha=1.0, theta=0.0, gamma=0.0)
    with pytest.raises(_ffi.NoConvergence, match="max_iter"):
        api.ncuts_labels(g, g.n, 0.075, max_iter=5)
    lab, ng, st = api.ncuts_labels(g, g.n, 0.075)          # the context is still usable afterwards
    assert st["unconverged"] == 0 and ng >= 1
    g.free()


def test_batch_accepts_an_unsplittable_chunk_anywhere(api):
    """A chunk too small to be split (n <= 2, or below split_lim of its own original size) may sit at any position
    of a batched call: the library orders the frontier itself (normalized_cut.py:39-40 gate per chunk)."""
    from autoinst_amd import synth
    big = [synth.synthetic_chunk(n, seed, tarl=True) for n, seed in ((6000, 1), (5000, 2))]
    tiny = np.array([[0.0, 0.0, 0.0], [0.2, 0.0, 0.0]])
    g_big = [api.build_affinity(c["points"], c["tarl"], alpha=1.0, theta=0.5, gamma=0.0) for c in big]
    g_tiny = api.build_affinity(tiny, None, alpha=1.0, theta=0.0, gamma=0.0)
    single = [api.ncuts_labels(g, g.n, 0.03) for g in g_big]
    graphs = [g_big[0], g_tiny, g_big[1]]
    labs, ngs, st = api.ncuts_labels_batch(graphs, [g_big[0].n, 2, 10 ** 9], 0.03)
    assert ngs[1] == 1 and np.all(labs[1] == 0)
    assert ngs[0] == single[0][1] and np.array_equal(labs[0], single[0][0])
    assert ngs[2] == 1 and np.all(labs[2] == 0)          # below split_lim of its (huge) original size
    labs, ngs, _ = api.ncuts_labels_batch(graphs, None, 0.03)
    assert np.array_equal(labs[2], single[1][0]) and ngs[2] == single[1][1] and ngs[1] == 1


def test_graph_ownership_across_contexts(api):
    """A graph's buffers belong to the context that built it: freeing it through another context is safe, and a
    graph that outlives its context fails cleanly instead of touching freed memory."""
    from autoinst_amd import _ffi, synth
    import ctypes as C
    lib = _ffi.load()
    ch = synth.synthetic_chunk(3000, seed=4, tarl=False, extent=14.0)
    a, b = api.Context(0), api.Context(0)
    g = api.build_affinity(ch["points"], None, alpha=1.0, theta=0.0, gamma=0.0, ctx=a)
    ref = g.to_scipy()
    assert lib.ai_csr_free(b._h, g._h) == 0     # foreign context: goes back to a's cache all the same
    g._h = None
    g2 = api.build_affinity(ch["points"], None, alpha=1.0, theta=0.0, gamma=0.0, ctx=a)   # re-uses the cached buffers
    assert abs(g2.to_scipy() - ref).max() == 0.0
    a.close()                                   # g2 outlives its context
    lab = np.empty(g2.n, dtype=np.int32)
    ng = C.c_int32()
    rc = lib.ai_ncut(b._h, g2._h, g2.n, 0.03, 0.01, None, lab.ctypes.data, C.byref(ng), None)
    assert rc == -1 and b"context was destroyed" in lib.ai_last_error()   # AI_ERR_BAD_ARG
    g2.free()
    b.close()


def test_apply_camera_on_an_uploaded_graph(api):
    """ai_affinity_apply_camera on a graph that came from ai_csr_from_host (no permutation of its own)."""
    from autoinst_amd import _ffi
    import ctypes as C
    rng = np.random.default_rng(8)
    pts = rng.normal(0, 1.0, (300, 3))
    dino = rng.normal(0, 1, (300, 384))
    A = ncuts_ref.affinity_sparse(pts, None, alpha=1.0)
    want = ncuts_ref.affinity_sparse(pts, None, [dino], alpha=1.0, gamma=0.07)
    g = api.DeviceGraph.from_scipy(A)
    d = np.ascontiguousarray(dino)
    _ffi.check(_ffi.load().ai_affinity_apply_camera(g.ctx._h, g._h, d.ctypes.data, 384, None, 0, 0.0, 0.07, _ffi.AI_MEM_HOST), "apply_camera")
    got = g.to_scipy()
    assert np.array_equal(got.indices, want.indices) and (np.abs(got.data - want.data) / want.data).max() <= 1e-12



def test_thousands_of_isolated_points_and_small_components(api):
    """A disconnected segment is split into ALL its components at once: 2 000 isolated points, 300 pairs and one blob of
    1 500 points come out as 2 000 + 300 + (the blob's own groups) groups; the model agrees group for group."""
    rng = np.random.default_rng(3)
    gx, gy = np.meshgrid(np.arange(50), np.arange(40))
    iso = np.stack([gx.ravel() * 2.5, gy.ravel() * 2.5, np.zeros(gx.size)], 1)                    # 2 000 singletons
    pa = np.stack([np.arange(300) * 2.5, np.full(300, -10.0), np.zeros(300)], 1)
    pairs = np.concatenate([pa, pa + np.array([0.4, 0.0, 0.0])])                                   # 300 two-point components
    blob = rng.normal(0.0, 1.5, (1500, 3)) + np.array([-30.0, -30.0, 0.0])
    pts = np.concatenate([iso, pairs, blob])[rng.permutation(2000 + 600 + 1500)]
    n = pts.shape[0]
    A = api.get_affinity_matrix(pts, alpha=1.0, theta=0.0, gamma=0.0)
    ncomp, comp = connected_components(A, directed=False)
    groups = api.ncuts(pts, alpha=1.0, theta=0.0, gamma=0.0, T=0.075)
    st = api.last_stats()
    lab = ncuts_ref.groups_to_labels(groups, n)
    assert (lab >= 0).all() and st["unconverged"] == 0
    # no group straddles two components, and every component of at most 1 % of the points (41) is exactly one group
    sizes = np.bincount(comp)
    for g in groups:
        assert np.unique(comp[g]).size == 1
    small = np.flatnonzero(sizes <= 0.01 * n)
    assert small.size >= 2300 and len(groups) >= small.size + 1
    exp = gpu_model.normalized_cut_model(A, n, np.arange(n), T=0.075)
    assert len(exp) == len(groups) and ncuts_ref.partitions_equal(lab, ncuts_ref.groups_to_labels(exp, n))


def test_run_chunks_equals_one_chunk_at_a_time(api):
    """`sharding.run_chunks` (host threads taking batches from one queue, one batched call per batch -- the chunk loop of
    run_pipeline.py:160-179 for one GPU) returns, chunk by chunk, the labels of a plain sequential call; chunks of very
    different sizes, one too small to be split."""
    from autoinst_amd import sharding, synth
    specs = [(9000, 31), (3000, 32), (15000, 33), (150, 34), (6000, 35), (12000, 36), (4000, 37), (7000, 38), (2500, 39)]
    chunks = [synth.synthetic_chunk(n, seed, tarl=True) for n, seed in specs]
    args = [(c["points"], c["tarl"]) for c in chunks]
    got = sharding.run_chunks(args, threads=3, batch=2, alpha=1.0, theta=0.5, gamma=0.0, T=0.03)
    assert len(got) == len(chunks)
    for c, lab in zip(chunks, got):
        g = api.build_affinity(c["points"], c["tarl"], alpha=1.0, theta=0.5, gamma=0.0)
        ref, _, _ = api.ncuts_labels(g, g.n, 0.03)
        g.free()
        assert np.array_equal(lab, ref)
    # one thread, one chunk per call: the same again
    again = sharding.run_chunks(args[:3], threads=1, batch=1, alpha=1.0, theta=0.5, gamma=0.0, T=0.03)
    assert all(np.array_equal(a, b) for a, b in zip(again, got[:3]))
    # the graphs of the next batches are built by builder threads on their own contexts (default: one) and cut on the worker's context:
    # none, and two, give the same labels
    for nb in (0, 2):
        other = sharding.run_chunks(args, threads=2, batch=3, builders=nb, alpha=1.0, theta=0.5, gamma=0.0, T=0.03)
        assert all(np.array_equal(a, b) for a, b in zip(other, got)), nb


def test_device_tensors_in_a_torch_first_process(tmp_path):
    """Inputs already resident in HBM (torch tensors -> raw device pointers, AI_MEM_DEVICE) give the labels of the host-array
    path.  torch ships its own HIP runtime and must initialise first (autoinst_amd/_ffi.py does that when torch is already
    imported), so this runs in a process of its own that imports torch before the package."""
    import subprocess
    import sys
    from conftest import ROOT
    code = f"""
import sys, numpy as np, torch
sys.path.insert(0, {ROOT!r})
from autoinst_amd import ncuts_api as api, sharding, synth
ch = [synth.synthetic_chunk(n, s, tarl=True) for n, s in ((8000, 41), (5000, 42), (11000, 43))]
host = [(c["points"], c["tarl"]) for c in ch]
dev = [(torch.from_numpy(p).cuda(), torch.from_numpy(f).cuda()) for p, f in host]
a = sharding.run_chunks(host, threads=2, batch=2, alpha=1.0, theta=0.5, gamma=0.0, T=0.03)
b = sharding.run_chunks(dev, threads=2, batch=2, alpha=1.0, theta=0.5, gamma=0.0, T=0.03)
assert all(np.array_equal(x, y) for x, y in zip(a, b))
g = api.build_affinity(dev[0][0], dev[0][1], alpha=1.0, theta=0.5, gamma=0.0)
lab, ng, _ = api.ncuts_labels(g, g.n, 0.03); g.free()
assert np.array_equal(lab, a[0]) and ng > 1
print("ok", ng)
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().startswith("ok"), r.stdout + r.stderr
