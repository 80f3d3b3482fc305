"""Seeded synthetic "surface chunk" generator.

The reference ships no data and its sample map is not reachable offline, so every
benchmark and parity test runs on chunks made here.  A chunk imitates what
``chunk_generation.chunks_from_pointcloud`` hands to ``ncuts_chunk``
(reference ``pipeline/utils/point_cloud/chunk_generation.py:96-180``): points of
object surfaces, voxel-downsampled at ``MAJOR_VOXEL_SIZE`` = 0.35 m
(``pipeline/config.py:56``), so that the 1.0 m radius graph
(``PROXIMITY_THRESHOLD``, ``pipeline/config.py:65``) has ~28-37 neighbours per point.

Recipe: boxes with centre ~U(-X/2, X/2)^2 x U(0, 2) m, size ~U(0.5, 4)^3 m, 200-3000
points on the 6 faces; first point per 0.35 m voxel kept, truncated to N.  The box
index (+1) is the ground-truth instance id used by the AP / S_assoc scorer.
"""
from __future__ import annotations

import numpy as np

VOXEL = 0.35

# extent (metres) that keeps the neighbour count of real 0.35 m chunks, per N
_EXTENTS = {10_000: 25.0, 50_000: 60.0, 200_000: 120.0, 1_000_000: 270.0}


def extent_for(n: int) -> float:
    """Square side X for an N-point chunk (25 m @10k ... 270 m @1M)."""
    if n in _EXTENTS:
        return _EXTENTS[n]
    # same areal density as the 10k / 25 m chunk
    return float(25.0 * np.sqrt(n / 10_000.0))


def _box_surface_points(rng, centre, size, m):
    """m points uniformly on the 6 faces of an axis-aligned box."""
    face = rng.integers(0, 6, size=m)
    u = rng.random((m, 3)) - 0.5
    axis = face // 2
    side = (face % 2).astype(np.float64) - 0.5
    u[np.arange(m), axis] = side
    return centre[None, :] + u * size[None, :]


def surface_chunk(n: int, seed: int = 0, extent: float | None = None):
    """Return (points[n,3] f64, gt_instance[n] int64).

    Deterministic for (n, seed, extent).  Points keep generation order after the
    voxel de-duplication, i.e. they are NOT spatially sorted (like open3d's
    voxel_down_sample output, which is hash-ordered).
    """
    rng = np.random.default_rng(seed)
    X = extent_for(n) if extent is None else float(extent)
    pts_l, ids_l = [], []
    seen = np.zeros(0, dtype=np.int64)
    total = 0
    box = 0
    target = int(1.05 * n) + 1
    while total < target:
        # a batch of boxes per round keeps the python overhead negligible at 1M
        nb = max(8, (target - total) // 400)
        bp, bi = [], []
        for _ in range(nb):
            box += 1
            centre = np.array([rng.uniform(-X / 2, X / 2), rng.uniform(-X / 2, X / 2), rng.uniform(0.0, 2.0)])
            size = rng.uniform(0.5, 4.0, size=3)
            m = int(rng.integers(200, 3001))
            bp.append(_box_surface_points(rng, centre, size, m))
            bi.append(np.full(m, box, dtype=np.int64))
        p = np.concatenate(bp)
        i = np.concatenate(bi)
        key = _voxel_key(p)
        # first occurrence inside this batch, in generation order
        _, first = np.unique(key, return_index=True)
        first.sort()
        p, i, key = p[first], i[first], key[first]
        # drop voxels already occupied by earlier batches
        fresh = ~np.isin(key, seen, assume_unique=False)
        p, i, key = p[fresh], i[fresh], key[fresh]
        seen = np.concatenate([seen, key])
        pts_l.append(p)
        ids_l.append(i)
        total += p.shape[0]
    pts = np.concatenate(pts_l)[:n]
    ids = np.concatenate(ids_l)[:n]
    return np.ascontiguousarray(pts, dtype=np.float64), ids


def _voxel_key(p):
    v = np.floor(p / VOXEL).astype(np.int64)
    v -= np.array([-(1 << 19), -(1 << 19), -(1 << 19)])
    return (v[:, 0] << 42) | (v[:, 1] << 21) | v[:, 2]


def surrogate_features(gt_ids, dim: int, seed: int = 0, zero_frac: float = 0.05, noise: float = 0.3):
    """TARL / DINO surrogate: f = noise*N(0,1)^dim + one-hot-ish box code.

    Stored the way the reference holds pooled features: float64, an all-zero row
    meaning "no feature for this point" (``chunk_generation.py:243-256`` yields a
    zero row when the 0.175 m radius search finds nothing).  Values are first rounded
    to float32, because the on-disk TARL / DINO features are float32
    (``kitti_odometry_dataset.py:251-281``).
    """
    rng = np.random.default_rng(seed + 7919 * dim)
    n = gt_ids.shape[0]
    f = noise * rng.standard_normal((n, dim)).astype(np.float32)
    code = (gt_ids * 2654435761 % dim).astype(np.int64)
    f[np.arange(n), code] += 1.0
    f[np.arange(n), (code * 7 + 3) % dim] += 1.0
    zero = rng.random(n) < zero_frac
    f[zero] = 0.0
    return f.astype(np.float64)


def synthetic_chunk(n: int, seed: int = 0, tarl: bool = True, dino: bool = False, extent: float | None = None):
    """Convenience: dict(points, gt, tarl|None, dino|None) for a cfg-named workload."""
    pts, gt = surface_chunk(n, seed, extent)
    out = {"points": pts, "gt": gt, "tarl": None, "dino": None}
    if tarl:
        out["tarl"] = surrogate_features(gt, 96, seed)
    if dino:
        out["dino"] = surrogate_features(gt, 384, seed + 1, zero_frac=0.05)
    return out
