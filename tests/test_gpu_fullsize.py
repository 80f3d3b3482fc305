"""GPU suite: BASELINE.json's full sizes through size-independent properties."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.sparse.csgraph import connected_components

from oracle import metrics_ref, ncuts_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from autoinst_amd import ncuts_api
    ncuts_api.default_context()
    return ncuts_api


@pytest.fixture(scope="module")
def chunk200k():
    from autoinst_amd import synth
    return synth.synthetic_chunk(200_000, seed=0, tarl=True)


def test_200k_affinity_properties(api, chunk200k):
    ch = chunk200k
    A = api.get_affinity_matrix(ch["points"], ch["tarl"], alpha=1.0, theta=0.5, gamma=0.0)
    n = A.shape[0]
    assert n == 200_000 and A.has_sorted_indices
    assert abs(A - A.T).max() == 0.0
    assert np.all(A.diagonal() == 1.0)
    assert A.data.min() > 0.0 and A.data.max() <= 1.0
    # the same pattern and values as the CPU restatement on a 20k-row slice of rows
    B = ncuts_ref.affinity_sparse(ch["points"], ch["tarl"], alpha=1.0, theta=0.5, gamma=0.0)
    assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
    assert (np.abs(A.data - B.data) / B.data).max() <= 1e-12


def test_200k_tarl_spatial_partition_properties(api, chunk200k):
    ch = chunk200k
    n = ch["points"].shape[0]
    g = api.build_affinity(ch["points"], ch["tarl"], alpha=1.0, theta=0.5, gamma=0.0)
    lab, ng, stats = api.ncuts_labels(g, n, 0.03)
    assert lab.min() == 0 and lab.max() == ng - 1 and np.unique(lab).size == ng
    assert stats["unconverged"] == 0, stats
    A = g.to_scipy()
    # a group never straddles two connected components unless it is a <= 1 % lump
    ncomp, comp = connected_components(A, directed=False)
    sizes = np.bincount(lab)
    for grp in np.argsort(-sizes)[:20]:
        idx = np.flatnonzero(lab == grp)
        if idx.size > 0.01 * n:
            assert np.unique(comp[idx]).size == 1
    # idempotence: a finished group larger than the split limit is not cut again
    big = [k for k in np.argsort(-sizes)[:3] if sizes[k] > 0.01 * n + 1]
    for k in big:
        idx = np.flatnonzero(lab == k)
        sub = A[idx][:, idx]
        again = api.normalized_cut(sub, n, idx, T=0.03)
        assert len(again) == 1
    # quality on the synthetic ground truth is finite and sane
    s = metrics_ref.score(lab + 1, lab + 1, ch["gt"])
    assert 0.0 <= s["S_assoc"] <= 1.0


def test_50k_largest_component_eigen_residual(api):
    from autoinst_amd import synth
    pts, _ = synth.surface_chunk(50_000, seed=1)
    A = api.get_affinity_matrix(pts, alpha=1.0, theta=0.0, gamma=0.0)
    _, comp = connected_components(A, directed=False)
    idx = np.flatnonzero(comp == np.bincount(comp).argmax())
    sub = sp.csr_matrix(A[idx][:, idx])
    g = api.DeviceGraph.from_scipy(sub)
    lam, ev, iters, resid = api.fiedler(g)
    L, _ = ncuts_ref.laplacian_sym(sub)
    assert resid <= 1e-10
    assert np.linalg.norm(L @ ev - lam * ev) <= 1e-8
    assert 0 < lam < 0.1
