"""ORACLE -- test infrastructure.  NumPy / SciPy restatement of the two point-cloud steps next to the
hot path: ``chunk_generation.py:243-256`` (radius-mean TARL pooling) and
``point_cloud_utils.py:144-174`` (1-NN re-projection).  The reference runs them on open3d's
KDTreeFlann, absent offline; cKDTree answers the same queries (radius search strict, as nanoflann's).
Parity unpinned by the reference (it has no tests); pinned by brute force on small inputs in tests/.
"""
import numpy as np
from scipy.spatial import cKDTree


def tarl_pool(points_major, tarl_points, tarl_features, radius=0.175):
    pts = np.asarray(points_major, dtype=np.float64)
    src = np.asarray(tarl_points, dtype=np.float64)
    feats = np.asarray(tarl_features).astype(np.float64)       # np.concatenate with a float64 seed, :240-242
    out = np.zeros((pts.shape[0], feats.shape[1]))             # :245
    if src.shape[0] == 0:
        return out
    tree = cKDTree(src)
    for i, idx in enumerate(tree.query_ball_point(pts, radius)):
        idx = np.asarray(idx, dtype=np.int64)
        if idx.size:
            d = np.linalg.norm(src[idx] - pts[i], axis=1)
            idx = idx[d < radius]                              # nanoflann: dist < radius
        if idx.size:
            out[i] = np.mean(feats[idx], axis=0)               # :251-252
    return out


def nn1_reproject(features_to, points_to, features_from, points_from, max_radius=None, no_feature_label=(1, 0, 0)):
    features_to = np.array(features_to, copy=True)
    d, idx = cKDTree(np.asarray(points_from)).query(np.asarray(points_to), k=1)
    features_to[:] = np.asarray(features_from)[idx]
    if max_radius is not None:
        features_to[d > max_radius] = np.asarray(no_feature_label, dtype=features_to.dtype)
    return features_to
