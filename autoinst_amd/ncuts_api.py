"""Host-side mirror of the reference's NCuts interface, backed by ``libautoinst_hip.so``.

Reference surface (paths relative to the reference root):

* ``normalized_cut(w, num_points_orig, labels, T=0.01, split_lim=0.01)`` --
  ``pipeline/ncuts/normalized_cut.py:37-63``: same arguments, same return type (a list of
  ``labels`` slices, groups in the recursion's emission order, members ascending);
* ``ncuts_chunk(...)`` -- ``pipeline/ncuts/ncuts_utils.py:28-204``: same signature and 5-tuple;
  its lines 60-174 (affinity + normalized cut) become one call into this module, the open3d
  glue stays as the reference wrote it and needs the reference's own ``utils`` package;
* ``get_affinity_matrix`` / ``ncuts`` -- names from the project brief that do NOT exist in the
  reference at this commit; provided as the array-level entries.

Nothing here computes on the CPU: every numeric step is a HIP kernel.  Without the shared
library or without a gfx950 device the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import scipy.sparse as sp

from . import _ffi
from .config import CONFIG, PROXIMITY_THRESHOLD, SPLIT_LIM

__all__ = ["Context", "DeviceGraph", "get_affinity_matrix", "build_affinity", "normalized_cut", "ncuts",
           "ncuts_labels", "ncuts_labels_batch", "ncuts_chunk", "default_context", "last_stats"]


class Context:
    """One main + one check HIP stream, workspace arena and graph-buffer cache on one GPU (``ai_ctx``).

    Not thread-safe: one per host thread (several per process are fine and are how one GPU is kept busy, see bench.py);
    ``device`` defaults to ``LOCAL_RANK``."""

    def __init__(self, device: int | None = None):
        lib = _ffi.load()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        h = C.c_void_p()
        _ffi.check(lib.ai_ctx_create(int(device), C.byref(h)), "ai_ctx_create")
        self._h = h
        self.device = int(device)

    def mem_info(self) -> dict:
        """Device memory the context holds between calls (``ai_ctx_mem_info``), in bytes."""
        out = (C.c_int64 * 4)()
        _ffi.check(_ffi.load().ai_ctx_mem_info(self._h, out), "ai_ctx_mem_info")
        return {"workspace": out[0], "workspace_blocks": out[1], "graphs_live": out[2], "graphs_kept": out[3]}

    def close(self):
        if getattr(self, "_h", None):
            _ffi.load().ai_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None
_last_stats: dict | None = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def last_stats() -> dict | None:
    """Counters and device timings of the most recent ``ai_ncut`` call of this process."""
    return _last_stats


class DeviceGraph:
    """Symmetric affinity graph resident in HBM (``ai_csr``)."""

    def __init__(self, ctx: Context, handle):
        self.ctx = ctx
        self._h = handle
        n, nnz = C.c_int64(), C.c_int64()
        _ffi.check(_ffi.load().ai_csr_dims(handle, C.byref(n), C.byref(nnz)), "ai_csr_dims")
        self.n, self.nnz = int(n.value), int(nnz.value)
        self.shape = (self.n, self.n)

    @classmethod
    def from_scipy(cls, w, ctx: Context | None = None) -> "DeviceGraph":
        ctx = ctx or default_context()
        w = sp.csr_matrix(w)
        if w.shape[0] != w.shape[1]:
            raise ValueError("affinity matrix must be square")
        indptr = np.ascontiguousarray(w.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(w.indices, dtype=np.int32)
        data = np.ascontiguousarray(w.data, dtype=np.float64)
        h = C.c_void_p()
        _ffi.check(_ffi.load().ai_csr_from_host(ctx._h, w.shape[0], indptr.ctypes.data, indices.ctypes.data,
                                                data.ctypes.data, C.byref(h)), "ai_csr_from_host")
        return cls(ctx, h)

    def to_scipy(self) -> sp.csr_matrix:
        """``scipy.sparse.csr_matrix(A)`` of ``ncuts_utils.py:167``: original order, sorted columns."""
        indptr = np.empty(self.n + 1, dtype=np.int64)
        indices = np.empty(self.nnz, dtype=np.int32)
        data = np.empty(self.nnz, dtype=np.float64)
        _ffi.check(_ffi.load().ai_csr_export(self.ctx._h, self._h, indptr.ctypes.data, indices.ctypes.data,
                                             data.ctypes.data), "ai_csr_export")
        return sp.csr_matrix((data, indices, indptr), shape=self.shape)

    def free(self):
        if getattr(self, "_h", None):
            _ffi.load().ai_csr_free(self.ctx._h, self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _as_f64(a, cols=None, name="array"):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim != 2 or (cols is not None and a.shape[1] != cols):
        raise ValueError(f"{name} must be 2-D" + (f" with {cols} columns" if cols else ""))
    return a


def _is_device_tensor(a):
    return a is not None and hasattr(a, "data_ptr") and bool(getattr(a, "is_cuda", False))


def _dev_f64(a, cols, name):
    """A torch tensor already resident in HBM: float64, contiguous, (n, cols)."""
    import torch
    if a.dtype != torch.float64 or not a.is_contiguous() or a.dim() != 2 or (cols is not None and a.shape[1] != cols):
        raise ValueError(f"{name} on the device must be a contiguous float64 (n, {cols or 'F'}) tensor")
    return a


def build_affinity(points, tarl=None, dino=None, *, alpha=CONFIG["alpha"], theta=CONFIG["theta"],
                   gamma=CONFIG["gamma"], radius=PROXIMITY_THRESHOLD, sam=None, beta=0.0,
                   ctx: Context | None = None) -> DeviceGraph:
    """Affinity graph of one chunk, left on the device (``ncuts_utils.py:60-67,112-167``).

    ``A_ij = 1[d_ij <= radius] * exp(-theta t_ij) * exp(-alpha d_ij) * exp(-beta s_ij) * exp(-gamma g_ij)``
    with the reference's rules: a falsy weight drops its factor, all-zero TARL rows have t = 0, A_ii = 1;
    ``s_ij`` = fraction of the views in which both points carry a SAM id and the ids differ (``sam``:
    (N, views) int ids of one camera, -1 = none; ``utils/image/image_utils.py:64-89``; beta is 0 in every
    shipped config).  Inputs are NumPy arrays (copied to the device by the call) or torch tensors that
    already live on the context's GPU (used in place; PyTorch here is only the owner of the HBM buffer).
    """
    ctx = ctx or default_context()
    # one matrix per camera (ncuts_utils.py:118-123, :128-133 loop over them); a bare array is one camera
    dino_list = list(dino) if isinstance(dino, (list, tuple)) else ([] if dino is None else [dino])
    sam_list = list(sam) if isinstance(sam, (list, tuple)) else ([] if sam is None else [sam])
    if gamma and not dino_list:
        raise ValueError("The length should be longer than 0!")  # ncuts_utils.py:126-127
    if beta and not sam_list:
        raise ValueError("The length should be longer than 0!")  # ncuts_utils.py:116-117
    dino = dino_list[0] if (gamma and dino_list) else None
    sam = sam_list[0] if (beta and sam_list) else None
    extra_dino = dino_list[1:] if gamma else []
    extra_sam = sam_list[1:] if beta else []
    if theta and tarl is None:
        raise ValueError("theta != 0 needs TARL features")
    on_dev = _is_device_tensor(points)
    if on_dev:
        import torch
        pts = _dev_f64(points, 3, "points")
        t = _dev_f64(tarl, None, "tarl") if theta else None
        d = _dev_f64(dino, None, "dino") if gamma else None
        for f in (t, d):
            if f is not None and not _is_device_tensor(f):
                raise ValueError("points are on the device, so features must be too")
        sm = None
        if beta:
            if not _is_device_tensor(sam) or sam.dtype != torch.int32 or not sam.is_contiguous() or sam.dim() != 2:
                raise ValueError("points are on the device, so sam must be a contiguous int32 (N, views) tensor there too")
            sm = sam
        torch.cuda.current_stream(pts.device).synchronize()  # producers of the buffers have finished
        ptr = lambda a: C.c_void_p(a.data_ptr()) if a is not None else None
        mem = _ffi.AI_MEM_DEVICE
    else:
        pts = _as_f64(points, 3, "points")
        t = _as_f64(tarl, None, "tarl") if theta else None
        d = _as_f64(dino, None, "dino") if gamma else None
        sm = None
        if beta:
            sm = np.ascontiguousarray(np.asarray(sam), dtype=np.int32)
            if sm.ndim != 2:
                raise ValueError("sam must be (N, views)")
        ptr = lambda a: a.ctypes.data if a is not None else None
        mem = _ffi.AI_MEM_HOST
    n = pts.shape[0]
    for f, nm in ((t, "tarl"), (d, "dino"), (sm, "sam")):
        if f is not None and f.shape[0] != n:
            raise ValueError(f"{nm} has {f.shape[0]} rows for {n} points")
    h = C.c_void_p()
    st = _ffi.load().ai_affinity_build_sam(
        ctx._h, ptr(pts), n, ptr(t), t.shape[1] if t is not None else 0, ptr(d), d.shape[1] if d is not None else 0,
        ptr(sm), sm.shape[1] if sm is not None else 0, float(alpha or 0.0), float(beta or 0.0), float(gamma or 0.0),
        float(theta or 0.0), float(radius), mem, C.byref(h))
    _ffi.check(st, "ai_affinity_build")
    graph = DeviceGraph(ctx, h)
    for c in range(max(len(extra_dino), len(extra_sam))):
        ed = extra_dino[c] if c < len(extra_dino) else None
        es = extra_sam[c] if c < len(extra_sam) else None
        if on_dev:
            import torch
            ed = _dev_f64(ed, None, "dino") if ed is not None else None
            if es is not None and (not _is_device_tensor(es) or es.dtype != torch.int32 or not es.is_contiguous()):
                raise ValueError("points are on the device, so sam must be a contiguous int32 tensor there too")
        else:
            ed = _as_f64(ed, None, "dino") if ed is not None else None
            es = np.ascontiguousarray(np.asarray(es), dtype=np.int32) if es is not None else None
        for f, nm in ((ed, "dino"), (es, "sam")):
            if f is not None and f.shape[0] != n:
                graph.free()
                raise ValueError(f"{nm} has {f.shape[0]} rows for {n} points")
        _ffi.check(_ffi.load().ai_affinity_apply_camera(ctx._h, graph._h, ptr(ed), ed.shape[1] if ed is not None else 0, ptr(es),
                                                        es.shape[1] if es is not None else 0, float(beta if es is not None else 0.0),
                                                        float(gamma if ed is not None else 0.0), mem), "ai_affinity_apply_camera")
    return graph


def get_affinity_matrix(points, tarl=None, dino=None, *, alpha=CONFIG["alpha"], theta=CONFIG["theta"],
                        gamma=CONFIG["gamma"], radius=PROXIMITY_THRESHOLD, sam=None, beta=0.0,
                        ctx: Context | None = None) -> sp.csr_matrix:
    """The CSR matrix the reference hands to ``normalized_cut`` (``ncuts_utils.py:167``)."""
    g = build_affinity(points, tarl, dino, alpha=alpha, theta=theta, gamma=gamma, radius=radius, sam=sam, beta=beta, ctx=ctx)
    try:
        return g.to_scipy()
    finally:
        g.free()


def _opts(tol, max_iter, check_every, time_spmv=False, window_rows=None):
    """``time_spmv``: every SpMV launch stamps its span on the device clock (fills ``ms_spmv``).
    ``window_rows``: points of the chunks of a batched call that iterate at one time (None: the library's default)."""
    flag = 2 if time_spmv else 0
    return _ffi.NcutOpts(float(tol or 0.0), int(max_iter or 0), int(check_every or 0), flag, int(window_rows or 0))


def ncuts_labels(graph: DeviceGraph, num_points_orig: int, T: float, split_lim: float = SPLIT_LIM, *,
                 tol=None, max_iter=None, check_every=None, time_spmv=False, ctx: Context | None = None):
    """Run the recursion on a device graph; returns (labels[int32 n], n_groups, stats dict).

    ``ctx``: the context (streams, workspace) the call runs on; default: the one that built the graph.  A context serves ONE thread at a
    time: a thread that cuts graphs another thread's context built passes its own."""
    global _last_stats
    lab = np.empty(graph.n, dtype=np.int32)
    ng = C.c_int32()
    stats = _ffi.NcutStats()
    o = _opts(tol, max_iter, check_every, time_spmv)
    status = _ffi.load().ai_ncut((ctx or graph.ctx)._h, graph._h, int(num_points_orig), float(T), float(split_lim),
                                 C.byref(o), lab.ctypes.data, C.byref(ng), C.byref(stats))
    _last_stats = stats.as_dict()   # filled on AI_ERR_NO_CONVERGENCE too
    _ffi.check(status, "ai_ncut")
    return lab, int(ng.value), _last_stats


def ncuts_labels_batch(graphs, num_points_orig=None, T=CONFIG["T"], split_lim=SPLIT_LIM, *, tol=None, max_iter=None,
                       check_every=None, time_spmv=False, window_rows=None, ctx: Context | None = None):
    """`ncuts_labels` for several independent chunks in ONE call (``ai_ncut_batch``).

    The connected segments of all chunks iterate in one pool and share every kernel launch, which is
    how a GPU is kept busy by a map's many chunks; chunks beyond ``window_rows`` points wait inside the
    call and are admitted as earlier ones finish.  Returns ([labels_c], [n_groups_c], stats);
    each chunk's result is what `ncuts_labels` gives for it alone.

    ``ctx``: the context the call runs on (default: the one that built the first graph).  A context serves ONE thread at a time, so a
    thread that cuts graphs a builder thread's context has built passes its own (`sharding.run_chunks`, `bench.py`).
    """
    global _last_stats
    graphs = list(graphs)
    if not graphs:
        return [], [], None
    ctx = ctx or graphs[0].ctx
    k = len(graphs)
    norig = [g.n for g in graphs] if num_points_orig is None else [int(x) for x in num_points_orig]
    labs = [np.empty(g.n, dtype=np.int32) for g in graphs]
    gh = (C.c_void_p * k)(*[g._h for g in graphs])
    lp = (C.c_void_p * k)(*[a.ctypes.data for a in labs])
    no = (C.c_int64 * k)(*norig)
    ng = (C.c_int32 * k)()
    stats = _ffi.NcutStats()
    o = _opts(tol, max_iter, check_every, time_spmv, window_rows)
    status = _ffi.load().ai_ncut_batch(ctx._h, gh, k, no, float(T), float(split_lim), C.byref(o), lp, ng, C.byref(stats))
    _last_stats = stats.as_dict()   # filled on AI_ERR_NO_CONVERGENCE too (`last_stats()` after the exception)
    _ffi.check(status, "ai_ncut_batch")
    return labs, [int(x) for x in ng], _last_stats


def _groups_from_labels(lab, ng, labels):
    order = np.argsort(lab, kind="stable")
    counts = np.bincount(lab, minlength=ng)
    return [labels[idx] for idx in np.split(order, np.cumsum(counts)[:-1])]


def normalized_cut(w, num_points_orig, labels, T=0.01, split_lim=0.01, *, ctx: Context | None = None,
                   tol=None, max_iter=None):
    """Drop-in for ``pipeline/ncuts/normalized_cut.py:37`` on the GPU.

    ``w``: SciPy sparse matrix (any format, float) or a `DeviceGraph`; ``labels``: the ids of
    w's rows.  Returns ``list[np.ndarray]`` -- a partition of ``labels``.
    """
    labels = np.asarray(labels)
    own = not isinstance(w, DeviceGraph)
    g = DeviceGraph.from_scipy(w, ctx) if own else w
    try:
        if labels.shape[0] != g.n:
            raise ValueError(f"labels has {labels.shape[0]} entries for a {g.n}-row matrix")
        lab, ng, _ = ncuts_labels(g, int(num_points_orig), T, split_lim, tol=tol, max_iter=max_iter)
    finally:
        if own:
            g.free()
    return _groups_from_labels(lab, ng, labels)


def ncuts(points, tarl=None, dino=None, *, alpha=CONFIG["alpha"], theta=CONFIG["theta"], gamma=CONFIG["gamma"],
          T=CONFIG["T"], split_lim=SPLIT_LIM, radius=PROXIMITY_THRESHOLD, sam=None, beta=0.0,
          ctx: Context | None = None, tol=None, max_iter=None):
    """Array-level ``ncuts_chunk`` lines 60-174: points (+features) -> list of index arrays."""
    g = build_affinity(points, tarl, dino, alpha=alpha, theta=theta, gamma=gamma, radius=radius, sam=sam, beta=beta, ctx=ctx)
    try:
        lab, ng, _ = ncuts_labels(g, g.n, T, split_lim, tol=tol, max_iter=max_iter)
    finally:
        g.free()
    return _groups_from_labels(lab, ng, np.arange(g.n))


def fiedler(graph: DeviceGraph, *, tol=None, max_iter=None):
    """Test hook: (lambda2, ev[n], iterations, residual) of the whole graph (``ai_fiedler``)."""
    ev = np.empty(graph.n, dtype=np.float64)
    lam, it, rs = C.c_double(), C.c_int32(), C.c_double()
    o = _opts(tol, max_iter, None)
    _ffi.check(_ffi.load().ai_fiedler(graph.ctx._h, graph._h, C.byref(o), C.byref(lam), ev.ctypes.data,
                                      C.byref(it), C.byref(rs)), "ai_fiedler")
    return float(lam.value), ev, int(it.value), float(rs.value)


def sweep(graph: DeviceGraph, ev):
    """Test hook: (costs[10], mask[n] bool, mcut) of ``get_min_ncut`` for a given ev (``ai_sweep``)."""
    ev = np.ascontiguousarray(ev, dtype=np.float64)
    costs = np.empty(_ffi.NUM_CUTS, dtype=np.float64)
    mask = np.empty(graph.n, dtype=np.uint8)
    mcut = C.c_double()
    _ffi.check(_ffi.load().ai_sweep(graph.ctx._h, graph._h, ev.ctypes.data, costs.ctypes.data, mask.ctypes.data,
                                    C.byref(mcut)), "ai_sweep")
    return costs, mask.astype(bool), float(mcut.value)


def lsym_apply(graph: DeviceGraph, x):
    """Test hook: ``L_sym @ x`` through the SpMV kernel (``ai_lsym_apply``)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    _ffi.check(_ffi.load().ai_lsym_apply(graph.ctx._h, graph._h, x.ctypes.data, y.ctypes.data), "ai_lsym_apply")
    return y


def eigs_smallest(graph: DeviceGraph, k: int, *, tol=None, max_iter=None, check_every=None):
    """k smallest eigenpairs of L_sym (``ai_eigs_smallest``): (evals[k], evecs[n, k], steps, max residual)."""
    evals = np.empty(k, dtype=np.float64)
    vecs = np.empty((k, graph.n), dtype=np.float64)
    it, mr = C.c_int32(), C.c_double()
    o = _opts(tol, max_iter, check_every)
    _ffi.check(_ffi.load().ai_eigs_smallest(graph.ctx._h, graph._h, int(k), C.byref(o), evals.ctypes.data, vecs.ctypes.data,
                                            C.byref(it), C.byref(mr)), "ai_eigs_smallest")
    return evals, vecs.T, int(it.value), float(mr.value)


def bench_spmv(graph: DeviceGraph, reps=50):
    """(average kernel ms, algorithmic bytes per launch) of the fused Lanczos SpMV kernel."""
    ms, by = C.c_double(), C.c_double()
    _ffi.check(_ffi.load().ai_bench_spmv(graph.ctx._h, graph._h, int(reps), C.byref(ms), C.byref(by)), "ai_bench_spmv")
    return float(ms.value), float(by.value)


def bench_copy(ctx: "Context", nbytes=1 << 30, reps=10):
    """GB/s (read + written) of a plain device-to-device copy with 16-byte accesses: the box's stream rate."""
    g = C.c_double()
    _ffi.check(_ffi.load().ai_bench_copy(ctx._h, int(nbytes), int(reps), C.byref(g)), "ai_bench_copy")
    return float(g.value)


# --------------------------------------------------------------------------- reference call surface
def ncuts_chunk(dataset, chunk_downsample_dict, pcd_nonground_minor, T_pcd, sampled_indices_global,
                sequence=None, patchwise_indices=None):
    """Drop-in for ``pipeline/ncuts/ncuts_utils.py:28-204`` (same arguments, same 5-tuple).

    Must be imported from inside the reference's ``pipeline/`` tree (it uses the reference's own
    ``utils`` / ``config`` modules and open3d for everything that is not the hot path: feature
    pooling upstream, colour painting and ground handling downstream).  Lines 60-174 of the
    reference -- the dense affinity matrices, ``remove_isolated_points`` and ``normalized_cut`` --
    are replaced by `build_affinity` + `ncuts_labels`, the TARL radius-mean pooling (:136-142 ->
    chunk_generation.py:243-256) by `points_api.tarl_features_per_patch`, and the 1-NN colour
    re-projection (:185-188) by `points_api.nn1_reproject`.
    """
    import open3d as o3d  # noqa: F401  (reference dependency, not present in the build containers)
    import config as refcfg
    from utils.image.image_utils import dinov2_mean, image_based_features_per_patch
    from utils.point_cloud.chunk_generation import get_indices_feature_reprojection
    from utils.point_cloud.point_cloud_utils import get_statistical_inlier_indices, get_subpcd, transform_pcd
    from utils.visualization_utils import generate_random_colors

    cfg = refcfg.CONFIG
    first_id = patchwise_indices[sequence][0]
    center_id = chunk_downsample_dict["center_ids"][sequence]
    center_position = chunk_downsample_dict["center_positions"][sequence]
    chunk_indices = chunk_downsample_dict["indices"][sequence]
    cam_indices_global, _ = get_indices_feature_reprojection(sampled_indices_global, first_id,
                                                             adjacent_frames=refcfg.ADJACENT_FRAMES_CAM)
    tarl_indices_global, _ = get_indices_feature_reprojection(sampled_indices_global, center_id,
                                                              adjacent_frames=refcfg.ADJACENT_FRAMES_TARL)
    pcd_chunk = chunk_downsample_dict["pcd_nonground_chunks"][sequence]
    pcd_ground_chunk = chunk_downsample_dict["pcd_ground_chunks"][sequence]
    chunk_major = chunk_downsample_dict["pcd_nonground_chunks_major_downsampling"][sequence]
    points_major = np.asarray(chunk_major.points)
    num_points_major = points_major.shape[0]

    dino = sam = None
    sam_list, point2dino_list = [], []
    # the three call forms of ncuts_utils.py:69-110 (their return shapes differ)
    if cfg["beta"] and not cfg["gamma"]:
        sam_list = image_based_features_per_patch(dataset, pcd_nonground_minor, chunk_indices, chunk_major, T_pcd, cam_indices_global,
                                                  sam=True, dino=False)
    elif cfg["gamma"] and not cfg["beta"]:
        point2dino_list, _ = image_based_features_per_patch(dataset, pcd_nonground_minor, chunk_indices, chunk_major, T_pcd,
                                                            cam_indices_global, sam=False, dino=True, pcd_chunk=pcd_chunk)
    elif cfg["beta"] and cfg["gamma"]:
        sam_list, point2dino_list = image_based_features_per_patch(dataset, pcd_nonground_minor, chunk_indices, chunk_major, T_pcd,
                                                                   cam_indices_global, sam=True, dino=True)
    if cfg["beta"]:
        if len(sam_list) == 0:
            raise ValueError("The length should be longer than 0!")
        sam = [np.asarray(x) for x in sam_list]
    if cfg["gamma"]:
        feats = [dinov2_mean(p2d) for p2d in point2dino_list]
        if len(feats) == 0:
            raise ValueError("The length should be longer than 0!")
        dino = feats
    tarl = None
    if cfg["theta"]:
        # tarl_features_per_patch (chunk_generation.py:205-258): its per-point radius search and mean run on the device
        from .points_api import tarl_features_per_patch
        tarl = tarl_features_per_patch(dataset, chunk_major, T_pcd, center_position, tarl_indices_global,
                                       chunk_size=refcfg.CHUNK_SIZE, major_voxel_size=refcfg.MAJOR_VOXEL_SIZE,
                                       tarl_norm=refcfg.TARL_NORM, transform_pcd=transform_pcd)

    graph = build_affinity(points_major, tarl, dino, alpha=cfg["alpha"], theta=cfg["theta"], gamma=cfg["gamma"],
                           radius=refcfg.PROXIMITY_THRESHOLD, sam=sam, beta=cfg["beta"])
    try:
        lab, ng, _ = ncuts_labels(graph, num_points_major, cfg["T"], refcfg.SPLIT_LIM)
    finally:
        graph.free()
    grouped_labels = _groups_from_labels(lab, ng, np.arange(num_points_major))

    random_colors = generate_random_colors(600)
    pcd_color = np.zeros((num_points_major, 3))
    for i, s in enumerate(grouped_labels):
        pcd_color[s] = np.array(random_colors[i]) / 255
    pcd_chunk.paint_uniform_color([0, 0, 0])
    # kDTree_1NN_feature_reprojection (point_cloud_utils.py:144-174, a Python loop over the fine points) on the device
    from .points_api import nn1_reproject
    colors = nn1_reproject(np.asarray(pcd_chunk.colors), np.asarray(pcd_chunk.points), pcd_color, points_major)
    pcd_chunk.colors = o3d.utility.Vector3dVector(colors)

    inliers = get_statistical_inlier_indices(pcd_ground_chunk)
    ground_inliers = get_subpcd(pcd_ground_chunk, inliers)
    mean_hight = np.mean(np.asarray(ground_inliers.points)[:, 2])
    in_idcs = np.where(np.asarray(ground_inliers.points)[:, 2] < (mean_hight + refcfg.MEAN_HEIGHT))[0]
    cut_hight = get_subpcd(ground_inliers, in_idcs)
    cut_hight.paint_uniform_color([0, 0, 0])
    merged_chunk = pcd_chunk + cut_hight
    inst_ground = chunk_downsample_dict["kitti_labels"]["ground"]["instance"][sequence][inliers][in_idcs]
    seg_ground = chunk_downsample_dict["kitti_labels"]["ground"]["semantic"][sequence][inliers][in_idcs]
    return merged_chunk, pcd_chunk, cut_hight, inst_ground, seg_ground
