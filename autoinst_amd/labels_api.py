"""Host-side mirror of the scorer and of the chunk merge.

The per-point passes run on the device (``csrc/ai_labels.hip``): one contingency table of two label
arrays (`label_pairs`), the crop / bounding-box / shared-coordinate counts of one merge iteration
(`merge_associate`) and duplicate-point removal (`unique_points`).  What is left on the host is
arithmetic over instances, a few hundred numbers per map, written here in the reference's order so
that results are bit-equal:

* `Metrics` / `score` -- ``pipeline/metrics/metrics_class.py`` (``update_stats:137-179``,
  ``filter_labels:302-309``, ``get_tp_fp:60-114``, ``average_precision:181-235``,
  ``calculate_full_stats:315-340``) and ``pipeline/metrics/modified_LSTQ.py:23-80``;
* `merge_chunks_unite_instances2` -- ``pipeline/utils/point_cloud/point_cloud_utils.py:387-491``.

There is no CPU fallback: every function raises when the HIP library or a gfx950 device is missing.
"""
from __future__ import annotations

import numpy as np

from . import _ffi
from .ncuts_api import Context, default_context

OVERLAPS = [0.25, 0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95]   # metrics_class.py:37
AP_OVERLAPS = OVERLAPS[1:]                                                  # :38


def _as_i32(x, what):
    a = np.asarray(x)
    if a.ndim != 1:
        raise ValueError(f"{what} must be one-dimensional")
    if a.dtype != np.int32:
        if a.size and (a.min() < -2 ** 31 or a.max() >= 2 ** 31):
            raise ValueError(f"{what} does not fit in int32")
        a = a.astype(np.int32)
    return np.ascontiguousarray(a)


def label_pairs(a, b, *, ctx: Context | None = None):
    """Distinct ``(a[i], b[i])`` pairs in ascending order with their counts: (pa, pb, count)."""
    ctx = ctx or default_context()
    a, b = _as_i32(a, "a"), _as_i32(b, "b")
    if a.shape != b.shape:
        raise ValueError("label arrays differ in length")
    lib = _ffi.load()
    cap = 1 << 16
    while True:
        pa, pb = np.empty(cap, np.int32), np.empty(cap, np.int32)
        cnt = np.empty(cap, np.int64)
        total = _ffi.C.c_int64(0)
        _ffi.check(lib.ai_label_pairs(ctx._h, a.ctypes.data, b.ctypes.data, a.shape[0], _ffi.AI_MEM_HOST, cap, pa.ctypes.data,
                                      pb.ctypes.data, cnt.ctypes.data, _ffi.C.byref(total)), "ai_label_pairs")
        if total.value <= cap:
            k = total.value
            return pa[:k].copy(), pb[:k].copy(), cnt[:k].copy()
        cap = int(total.value)


class _Table:
    """Dense contingency table of (row label, gt label) with sorted label axes."""

    def __init__(self, pa, pb, cnt):
        self.rows, ri = np.unique(pa, return_inverse=True)
        self.cols, ci = np.unique(pb, return_inverse=True)
        self.m = np.zeros((self.rows.size, self.cols.size), dtype=np.int64)
        np.add.at(self.m, (ri, ci), cnt)

    def filtered(self, min_points):
        """``filter_labels`` (:302-309) applied to the row labels: instances with fewer than
        min_points points become background 0.  Returns a new table."""
        area = self.m.sum(1)
        small = area < min_points
        new_label = np.where(small, 0, self.rows)
        t = _Table.__new__(_Table)
        t.rows, ri = np.unique(new_label, return_inverse=True)
        t.cols = self.cols
        t.m = np.zeros((t.rows.size, self.cols.size), dtype=np.int64)
        np.add.at(t.m, ri, self.m)
        return t


def _iou(table):
    ap, ag = table.m.sum(1), table.m.sum(0)
    union = ap[:, None] + ag[None, :] - table.m
    return table.m / np.maximum(union, 1)       # intersection.size / union.size, :296-300


def _greedy(table, iou, thresh):
    """Matching order of ``get_tp_fp`` / ``average_precision``: predictions ascending, the first
    unused non-background gt (ascending) with IoU >= thresh."""
    used = np.zeros(table.cols.size, dtype=bool)
    gt_ok = table.cols != 0
    hits = []
    for a in range(table.rows.size):
        if table.rows[a] == 0:
            continue
        cand = np.flatnonzero((iou[a] >= thresh) & ~used & gt_ok)
        if cand.size:
            used[cand[0]] = True
            hits.append(True)
        else:
            hits.append(False)
    return hits


def _average_precision(table, iou, thresh):
    """``metrics_class.py:181-235`` with every confidence 0.5 (:193-195)."""
    precision, recall = [1.0], [0.0]
    tp = fp = 0
    fn = int((table.cols != 0).sum())
    for hit in _greedy(table, iou, thresh):
        if hit:
            tp += 1
            fn -= 1
        else:
            fp += 1
        precision.append(tp / float(tp + fp))
        recall.append(tp / float(tp + fn))
    trapz = getattr(np, "trapezoid", None) or np.trapz
    return float(trapz(precision, recall))


def _s_assoc(table, min_points):
    """``modified_LSTQ.py:23-80`` for one batch, from the (filtered all_labels, gt) table."""
    p_keep = (table.rows != 0) & (table.rows != -1)
    p_area = table.m.sum(1)
    g_area = table.m.sum(0)
    g_keep = (table.cols != 0) & (g_area > min_points)
    if not g_keep.any():
        return float("nan")
    outer = 0.0
    for g in np.flatnonzero(g_keep):
        ga = int(g_area[g])
        inner = 0.0
        if table.cols[g] > 0:
            for p in np.flatnonzero(p_keep & (table.rows > 0) & (table.m[:, g] > 0)):
                t = int(table.m[p, g])
                inner += t * (t / (ga + int(p_area[p]) - t))
        outer += float(inner) / float(ga)
    return outer / int(g_keep.sum())


def score(all_labels, pred_labels, gt_labels, min_points=200, *, ctx: Context | None = None):
    """One ``Metrics.update_stats`` on a fresh scorer: dict(p, r, f1, ap, ap0.25, ap0.5, S_assoc)."""
    pred = _Table(*label_pairs(pred_labels, gt_labels, ctx=ctx)).filtered(min_points)
    alll = pred if all_labels is pred_labels else _Table(*label_pairs(all_labels, gt_labels, ctx=ctx)).filtered(min_points)
    iou = _iou(pred)
    tps = int(sum(_greedy(pred, iou, 0.5)))
    n_pred = int((pred.rows != 0).sum())
    n_gt = pred.cols.size - 1 if (pred.cols == 0).any() else 0          # calculate_full_stats:323-325
    prec = tps / n_pred if n_pred else float("nan")
    rec = tps / n_gt if n_gt else float("nan")
    try:
        f1 = 2 * (prec * rec) / (prec + rec)
    except ZeroDivisionError:
        f1 = 0
    aps = {o: _average_precision(pred, iou, o) for o in OVERLAPS}
    ap = sum(aps[o] for o in AP_OVERLAPS) / float(len(AP_OVERLAPS))
    return {"p": prec, "r": rec, "f1": f1, "ap": ap, "ap0.25": aps[0.25], "ap0.5": aps[0.5],
            "S_assoc": _s_assoc(alll, min_points), "tps": tps, "n_pred": n_pred, "n_gt": n_gt}


class Metrics:
    """The reference's scorer object (``metrics_class.py:15``): same constructor arguments,
    ``update_stats`` signature, return value and ``sequence_metrics`` bookkeeping; precision /
    recall accumulate over calls like ``all_tp`` / ``all_pred_size`` / ``all_gt_size`` (:321-329)."""

    def __init__(self, name="NCuts", min_points=200, thresh=0.5, *, ctx: Context | None = None):
        self.name, self.min_points, self.thresh = name, min_points, thresh
        self._ctx = ctx
        self.all_tp = self.all_pred_size = self.all_gt_size = 0
        self.s_assoc_list = []
        self.ap = {}
        self.sequence_metrics = {'ap0.5': [], 'ap0.25': [], 'ap': [], 'p': [], 'r': [], 'f1': [], 'S_assoc': []}

    def update_stats(self, all_labels, pred_labels, gt_labels, confs=[], calc_all=True, calc_lstq=True):
        if len(confs):
            raise NotImplementedError("per-instance confidences are never passed by the reference pipeline")
        s = score(all_labels, pred_labels, gt_labels, self.min_points, ctx=self._ctx)
        self.all_tp += s["tps"]
        self.all_pred_size += s["n_pred"]
        self.all_gt_size += s["n_gt"]
        prec = self.all_tp / self.all_pred_size
        rec = self.all_tp / self.all_gt_size
        try:
            f1 = 2 * (prec * rec) / (prec + rec)
        except ZeroDivisionError:
            f1 = 0
        out = {"fScore": f1, "precision": prec, "recall": rec}
        self.s_assoc_list.append(s["S_assoc"])
        lstq = float(np.average(self.s_assoc_list))                      # modified_LSTQ.py:80
        self.ap = {o: s[k] for o, k in ((0.25, "ap0.25"), (0.5, "ap0.5"))}
        for k, v in (('p', prec), ('r', rec), ('f1', f1), ('ap0.25', s["ap0.25"]), ('ap0.5', s["ap0.5"]), ('ap', s["ap"]),
                     ('S_assoc', lstq)):
            self.sequence_metrics[k].append(v)
        return out, {"0.25": s["ap0.25"], "0.5": s["ap0.5"], "ap": s["ap"], "lstq": lstq}


# ----------------------------------------------------------------------------- chunk merge
def unique_points(points, *, ctx: Context | None = None):
    """Indices (ascending) of the first point of every distinct coordinate triple."""
    ctx = ctx or default_context()
    p = np.ascontiguousarray(points, dtype=np.float64)
    if p.ndim != 2 or p.shape[1] != 3:
        raise ValueError("points must be (N, 3)")
    keep = np.empty(p.shape[0], dtype=np.int32)
    n_keep = _ffi.C.c_int64(0)
    _ffi.check(_ffi.load().ai_unique_points(ctx._h, p.ctypes.data, p.shape[0], _ffi.AI_MEM_HOST, keep.ctypes.data,
                                            _ffi.C.byref(n_keep)), "ai_unique_points")
    return keep[: n_keep.value].copy()


def merge_associate(map_points, map_inst, chunk_points, chunk_inst, center, n_inst1, n_inst2, side_length=40.0, *,
                    ctx: Context | None = None):
    """Per-point counts of one merge iteration: dict(inter, common, n_scalars1, n_scalars2, n_points1)."""
    ctx = ctx or default_context()
    mp = np.ascontiguousarray(map_points, dtype=np.float64)
    cp = np.ascontiguousarray(chunk_points, dtype=np.float64)
    mi, ci = _as_i32(map_inst, "map_inst"), _as_i32(chunk_inst, "chunk_inst")
    if mp.ndim != 2 or mp.shape[1] != 3 or cp.ndim != 2 or cp.shape[1] != 3 or mi.shape[0] != mp.shape[0] or ci.shape[0] != cp.shape[0]:
        raise ValueError("points must be (N, 3) with one instance id per point")
    c = np.ascontiguousarray(center, dtype=np.float64)
    inter = np.empty((n_inst1, n_inst2), np.int32)
    common = np.empty((n_inst1, n_inst2), np.int32)
    ns1, ns2, np1 = np.empty(n_inst1, np.int32), np.empty(n_inst2, np.int32), np.empty(n_inst1, np.int32)
    _ffi.check(_ffi.load().ai_merge_associate(ctx._h, mp.ctypes.data, mi.ctypes.data, mp.shape[0], cp.ctypes.data, ci.ctypes.data,
                                              cp.shape[0], c.ctypes.data, float(side_length), n_inst1, n_inst2, _ffi.AI_MEM_HOST,
                                              inter.ctypes.data, common.ctypes.data, ns1.ctypes.data, ns2.ctypes.data, np1.ctypes.data),
               "ai_merge_associate")
    return {"inter": inter, "common": common, "n_scalars1": ns1, "n_scalars2": ns2, "n_points1": np1}


def _color_ids(colors):
    """(unique colours in np.unique(axis=0) order with black first if present, id per point);
    id 0 is reserved for black = street (:430, :438), instances are 1.. in lexicographic order."""
    c = np.ascontiguousarray(colors, dtype=np.float64)
    key = (c + 0.0).view(np.dtype((np.void, 24))).ravel()           # + 0.0 folds -0.0 into +0.0
    ukey, first, inv = np.unique(key, return_index=True, return_inverse=True)
    ucol = c[first]
    order = np.lexsort((ucol[:, 2], ucol[:, 1], ucol[:, 0]))          # numeric lexicographic, as np.unique(axis=0)
    black = np.all(ucol[order] == 0.0, axis=1)
    order = np.concatenate([order[black], order[~black]])
    rank = np.empty(order.size, dtype=np.int32)
    rank[order] = np.arange(order.size, dtype=np.int32) + (0 if black.any() else 1)
    table = np.zeros((order.size + (0 if black.any() else 1), 3))
    table[rank] = ucol
    return table, rank[inv.ravel()]


def _points_colors(chunk):
    if isinstance(chunk, (tuple, list)):
        return np.asarray(chunk[0], dtype=np.float64), np.asarray(chunk[1], dtype=np.float64)
    return np.asarray(chunk.points, dtype=np.float64), np.asarray(chunk.colors, dtype=np.float64)   # open3d-like


def merge_chunks_unite_instances2(chunks, icp=False, *, ctx: Context | None = None):
    """``point_cloud_utils.py:387-491`` on arrays: ``chunks`` is a list of ``(points, colors)`` pairs
    (or objects with ``.points`` / ``.colors``); returns the merged ``(points, colors)``.

    Instance identity is the colour, as in the reference.  Per new chunk the device does the crop,
    the boxes, the inside counts, the shared-coordinate counts and the duplicate removal; the
    association (:444-477) runs here over the few candidate pairs.
    """
    merge_p, merge_c = _points_colors(chunks[0])
    merge_p, merge_c = merge_p.copy(), merge_c.copy()
    for chunk in chunks[1:]:
        new_p, new_c = _points_colors(chunk)
        center = np.array([new_p[:, 0].mean(), new_p[:, 1].mean(), new_p[:, 2].mean()])      # :397-403
        table1, inst1 = _color_ids(merge_c)
        table2, inst2 = _color_ids(new_c)
        r = merge_associate(merge_p, inst1, new_p, inst2, center, table1.shape[0], table2.shape[0], 40.0, ctx=ctx)
        inter = r["inter"].astype(np.int64)
        union = r["n_scalars1"].astype(np.int64)[:, None] + r["n_scalars2"].astype(np.int64)[None, :] - r["common"]
        ids_chunk_1, ids_chunk_2, ious = [], [], []
        for id1, id2 in np.argwhere(inter > 0):                                               # row-major = the loops of :446-463
            iou = float(inter[id1, id2]) / float(union[id1, id2])
            if not iou > 0.01:
                continue
            if id2 not in ids_chunk_2:                                                        # :465-477
                ids_chunk_1.append(id1)
                ids_chunk_2.append(id2)
                ious.append(iou)
            else:
                i = ids_chunk_2.index(id2)
                if iou > ious[i]:
                    ious[i] = iou
                    ids_chunk_1[i] = id1
        colors_2 = new_c.copy()
        for id1, id2 in zip(ids_chunk_1, ids_chunk_2):                                        # :479-481
            colors_2[inst2 == id2] = table1[id1]
        merge_p = np.concatenate([merge_p, new_p])                                            # :488
        merge_c = np.concatenate([merge_c, colors_2])
        keep = unique_points(merge_p, ctx=ctx)                                                # :489
        merge_p, merge_c = merge_p[keep], merge_c[keep]
    return merge_p, merge_c
