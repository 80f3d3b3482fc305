"""On-disk formats either side of the NCuts hot path -- plain host I/O.

* TARL features: ``<scan>.bin`` = zlib-deflated raw float32 (N x 96)
  (written by ``Pointcloud-Models/tarl/tarl_extractor.py:84-89``, read by
  ``pipeline/dataset/kitti_odometry_dataset.py:251-281``);
* DINOv2 features: ``<frame>.npz`` with key ``feature_map`` (H' x W' x 384)
  (``kitti_odometry_dataset.py:224-249``);
* self-training sample: ``.npz`` with ``pts, ncut_labels, kitti_labels, cluster_labels, semantic``
  (``pipeline/dataset/dataset_utils.py:604-611``, read by
  ``self-training/mask_pls/datasets/pseudo_dataset.py:147-153``);
* per-chunk ``.pcd``: binary PCD with ``x y z rgb`` float32 fields, the instance encoded as the point's
  colour (``point_cloud_utils.py:65-75`` via open3d's writer).
"""
from __future__ import annotations

import zlib

import numpy as np

TARL_DIM = 96


def read_tarl_bin(path, dim: int = TARL_DIM) -> np.ndarray:
    with open(path, "rb") as f:
        raw = zlib.decompress(f.read())
    return np.frombuffer(raw, dtype=np.float32).reshape(-1, dim)


def write_tarl_bin(path, feats) -> None:
    data = np.ascontiguousarray(feats, dtype=np.float32)
    with open(path, "wb") as f:
        f.write(zlib.compress(data.tobytes()))


def read_dino_npz(path) -> np.ndarray:
    return np.load(path, allow_pickle=True)["feature_map"]


def write_selftrain_npz(path, pts, ncut_labels, kitti_labels, semantic) -> None:
    ncut_labels = np.asarray(ncut_labels)
    np.savez(path, pts=np.asarray(pts), ncut_labels=ncut_labels, kitti_labels=np.asarray(kitti_labels),
             cluster_labels=np.zeros_like(ncut_labels), semantic=np.asarray(semantic))


def read_selftrain_npz(path) -> dict:
    z = np.load(path)
    return {k: z[k] for k in ("pts", "ncut_labels", "kitti_labels", "cluster_labels", "semantic")}


def write_pcd_binary(path, points, colors) -> None:
    """Binary PCD as open3d writes a coloured cloud: float32 x, y, z and rgb packed into a float32."""
    pts = np.asarray(points, dtype=np.float32)
    rgb = np.clip(np.rint(np.asarray(colors, dtype=np.float64) * 255.0), 0, 255).astype(np.uint32)
    packed = ((rgb[:, 0] << 16) | (rgb[:, 1] << 8) | rgb[:, 2]).astype(np.uint32)
    rec = np.empty(pts.shape[0], dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgb", "<u4")])
    rec["x"], rec["y"], rec["z"], rec["rgb"] = pts[:, 0], pts[:, 1], pts[:, 2], packed
    n = pts.shape[0]
    header = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F F\n"
              f"COUNT 1 1 1 1\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n")
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(rec.tobytes())


def read_pcd_binary(path):
    """(points float64 (N,3), colors float64 (N,3) in [0,1]) of a binary ``x y z rgb`` PCD."""
    with open(path, "rb") as f:
        n = None
        fields = None
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("FIELDS"):
                fields = line.split()[1:]
            if line.startswith("POINTS"):
                n = int(line.split()[1])
            if line.startswith("DATA"):
                if line.split()[1] != "binary":
                    raise ValueError("only binary PCD files are supported")
                break
        if fields != ["x", "y", "z", "rgb"]:
            raise ValueError(f"unsupported PCD fields {fields}")
        rec = np.frombuffer(f.read(n * 16), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgb", "<u4")])
    pts = np.stack([rec["x"], rec["y"], rec["z"]], 1).astype(np.float64)
    rgb = rec["rgb"]
    col = np.stack([(rgb >> 16) & 255, (rgb >> 8) & 255, rgb & 255], 1).astype(np.float64) / 255.0
    return pts, col


def labels_to_colors(labels, seed: int = 0):
    """Instance id -> a distinct colour (label 0 = black background), the reference's label-as-colour
    encoding (``visualization_utils.py:25-33`` draws them unseeded; here they are reproducible)."""
    labels = np.asarray(labels).astype(np.int64)
    ids = np.unique(labels)
    rng = np.random.default_rng(seed)
    table = {}
    used = {(0, 0, 0)}
    for i in ids:
        if i == 0:
            table[i] = (0, 0, 0)
            continue
        while True:
            c = tuple(int(x) for x in rng.integers(0, 256, 3))
            if c not in used:
                used.add(c)
                table[i] = c
                break
    lut = np.array([table[i] for i in ids], dtype=np.float64) / 255.0
    return lut[np.searchsorted(ids, labels)]
