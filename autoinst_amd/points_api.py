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
from .config import CHUNK_SIZE, MAJOR_VOXEL_SIZE, TARL_NORM
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


def _transform_points(points, T):
    """open3d ``PointCloud.transform``: homogeneous 4x4 applied to every point, divided by w."""
    p = np.asarray(points, dtype=np.float64)
    T = np.asarray(T, dtype=np.float64)
    q = p @ T[:3, :3].T + T[:3, 3]
    w = p @ T[3, :3] + T[3, 3]
    return q / w[:, None]


def tarl_features_per_patch(dataset, pcd, T_pcd, center_position, tarl_indices, *, chunk_size=CHUNK_SIZE,
                            major_voxel_size=MAJOR_VOXEL_SIZE, tarl_norm=TARL_NORM, transform_pcd=None,
                            ctx: Context | None = None):
    """Drop-in for ``tarl_features_per_patch`` (``chunk_generation.py:205-258``, same five positional
    arguments): the scan loop (:222-241, dataset reads, pose transform, crop to the chunk box) stays
    on the host, the per-point radius search and mean (:243-256) run on the device.  ``dataset`` needs
    ``get_tarl_features`` / ``get_point_cloud`` / ``get_pose`` like the reference's; ``pcd`` needs
    ``.points``.  ``transform_pcd`` may be the reference's own function (open3d); the default applies
    the same homogeneous transform in NumPy.
    """
    tf = transform_pcd or _transform_points
    center_position = np.asarray(center_position, dtype=np.float64)
    max_position = center_position + 0.5 * np.asarray(chunk_size, dtype=np.float64)            # :219-220
    min_position = center_position - 0.5 * np.asarray(chunk_size, dtype=np.float64)
    pts_list, feat_list = [], []
    for points_index in tarl_indices:                                                            # :222
        tarl_features = np.asarray(dataset.get_tarl_features(points_index))
        coords = np.asarray(dataset.get_point_cloud(points_index))
        T_local2global = np.linalg.inv(T_pcd) @ dataset.get_pose(points_index)                   # :229-231
        coords = np.asarray(tf(coords, T_local2global))
        mask = np.where(np.all(coords > min_position, axis=1) & np.all(coords < max_position, axis=1))[0]   # :233-236
        pts_list.append(coords[mask])
        feat_list.append(np.asarray(tarl_features[mask], dtype=np.float32))
    points_major = np.asarray(pcd.points)
    if not pts_list:
        return np.zeros((points_major.shape[0], 96))                                              # :246 with no scan at all
    out = tarl_pool(points_major, np.concatenate(pts_list), np.concatenate(feat_list),
                    radius=major_voxel_size / 2.0, ctx=ctx)                                       # :243-252
    if tarl_norm:                                                                                # :253-254
        nrm = np.linalg.norm(out, axis=1)
        has = out.any(axis=1)
        out[has] /= nrm[has, None]
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
