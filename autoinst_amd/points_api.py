"""Host-side mirror of the two point-cloud steps around the NCuts hot path.

* `tarl_pool` -- the radius-mean pooling loop of ``tarl_features_per_patch``
  (``pipeline/utils/point_cloud/chunk_generation.py:243-256``);
* `nn1_reproject` -- ``kDTree_1NN_feature_reprojection``
  (``pipeline/utils/point_cloud/point_cloud_utils.py:144-174``).
Both run as HIP kernels over a uniform cell list; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi
from .config import MAJOR_VOXEL_SIZE
from .ncuts_api import Context, default_context


def tarl_pool(points_major, tarl_points, tarl_features, *, radius=MAJOR_VOXEL_SIZE / 2.0, ctx: Context | None = None):
    """(N, F) float64: mean TARL feature within `radius` of every major-voxel point, zero row if none.

    ``tarl_points`` (M, 3) are the concatenated, already transformed and cropped scan points and
    ``tarl_features`` (M, F) their float32 features (``chunk_generation.py:218-241`` stay in Python:
    they are dataset I/O).  The search is strict (< radius) like open3d's radius search.
    """
    ctx = ctx or default_context()
    q = np.ascontiguousarray(points_major, dtype=np.float64)
    s = np.ascontiguousarray(tarl_points, dtype=np.float64)
    f = np.ascontiguousarray(tarl_features, dtype=np.float32)
    if q.ndim != 2 or q.shape[1] != 3 or s.ndim != 2 or s.shape[1] != 3 or f.ndim != 2 or f.shape[0] != s.shape[0]:
        raise ValueError("points must be (N, 3) / (M, 3) and features (M, F)")
    out = np.zeros((q.shape[0], f.shape[1]), dtype=np.float64)
    if s.shape[0] == 0:
        return out  # tarl_features stays all-zero in the reference too
    cnt = np.empty(q.shape[0], dtype=np.int32)
    _ffi.check(_ffi.load().ai_radius_mean_pool(ctx._h, q.ctypes.data, q.shape[0], s.ctypes.data, s.shape[0], f.ctypes.data,
                                               f.shape[1], float(radius), _ffi.AI_MEM_HOST, out.ctypes.data, cnt.ctypes.data),
               "ai_radius_mean_pool")
    return out


def nn1_index(points_to, points_from, *, ctx: Context | None = None):
    """(index[Nt] int32, distance[Nt] float64) of the nearest `points_from` row for every `points_to` row."""
    ctx = ctx or default_context()
    t = np.ascontiguousarray(points_to, dtype=np.float64)
    f = np.ascontiguousarray(points_from, dtype=np.float64)
    idx = np.empty(t.shape[0], dtype=np.int32)
    dist = np.empty(t.shape[0], dtype=np.float64)
    _ffi.check(_ffi.load().ai_nn1_project(ctx._h, t.ctypes.data, t.shape[0], f.ctypes.data, f.shape[0], _ffi.AI_MEM_HOST,
                                          idx.ctypes.data, dist.ctypes.data), "ai_nn1_project")
    return idx, dist


def nn1_reproject(features_to, points_to, features_from, points_from, max_radius=None, no_feature_label=(1, 0, 0), *,
                  ctx: Context | None = None):
    """Drop-in arithmetic of ``kDTree_1NN_feature_reprojection`` on arrays (no open3d objects)."""
    features_to = np.array(features_to, copy=True)
    idx, dist = nn1_index(points_to, points_from, ctx=ctx)
    features_to[:] = np.asarray(features_from)[idx]
    if max_radius is not None:
        features_to[dist > max_radius] = np.asarray(no_feature_label, dtype=features_to.dtype)
    return features_to
