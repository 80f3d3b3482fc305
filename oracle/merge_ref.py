"""ORACLE -- test infrastructure.  NumPy restatement of the reference's chunk merge.

Follows ``pipeline/utils/point_cloud/point_cloud_utils.py:387-491``
(``merge_chunks_unite_instances2``) line by line on plain arrays: a "point cloud" here is a pair
``(points (N, 3) float64, colors (N, 3) float64)``.  The three open3d calls it makes are restated
from their documented behaviour: ``crop(AxisAlignedBoundingBox)`` keeps ``min <= p <= max`` in
the original order, ``+=`` concatenates, ``remove_duplicated_points()`` keeps the first point of
every distinct coordinate triple in the original order.

PARITY UNPINNED: the reference function needs open3d, which is not importable in the build
container, and the reference has no tests or fixtures for it; this file is checked only against
hand-derived cases (``tests/test_labels.py``).  The product never imports it.
"""
from __future__ import annotations

import numpy as np


def crop(points, colors, min_bound, max_bound):
    """open3d ``PointCloud.crop``: inclusive on both sides, order kept."""
    keep = np.all(points >= min_bound, axis=1) & np.all(points <= max_bound, axis=1)
    return points[keep], colors[keep]


def remove_duplicated_points(points, colors):
    """open3d ``remove_duplicated_points``: first occurrence of every coordinate triple, order kept."""
    if points.shape[0] == 0:
        return points, colors
    _, first = np.unique(points, axis=0, return_index=True)
    first.sort()
    return points[first], colors[first]


def merge_chunks_unite_instances2(chunks):
    """``point_cloud_utils.py:387-491``; ``chunks`` = list of (points, colors); returns (points, colors)."""
    merge_p = np.array(chunks[0][0], dtype=np.float64, copy=True)                       # :391-393
    merge_c = np.array(chunks[0][1], dtype=np.float64, copy=True)
    for new_p, new_c in chunks[1:]:                                                      # :395
        new_p = np.asarray(new_p, dtype=np.float64)
        x, y, z = new_p[:, 0].mean(), new_p[:, 1].mean(), new_p[:, 2].mean()             # :397-403
        side_length = 40                                                                  # :405
        center_point = np.array([x, y, z])
        half_side = side_length / 2.0
        points_1, colors_1 = crop(merge_p, merge_c, center_point - half_side, center_point + half_side)  # :406-417
        points_2 = new_p
        colors_2 = np.array(new_c, dtype=np.float64, copy=True)                          # :419-423
        unique_colors_1 = np.unique(colors_1, axis=0)                                    # :425-426
        unique_colors_2 = np.unique(colors_2, axis=0)
        instance2point_1 = {}
        for i in range(unique_colors_1.shape[0]):                                        # :428-434
            if not np.all(unique_colors_1[i] == 0.0):
                inds = np.where(np.all(colors_1 == unique_colors_1[i], axis=1))[0]
                instance2point_1[i] = {"points": points_1[inds], "inds": inds}
        instance2point_2 = {}
        for i in range(unique_colors_2.shape[0]):                                        # :436-442
            if not np.all(unique_colors_2[i] == 0.0):
                inds = np.where(np.all(colors_2 == unique_colors_2[i], axis=1))[0]
                instance2point_2[i] = {"points": points_2[inds], "inds": inds}
        id_pairs_iou = []
        for id_1, entries_1 in instance2point_1.items():                                 # :444-463
            points1 = entries_1["points"]
            min_bound = np.min(points1, axis=0)
            max_bound = np.max(points1, axis=0)
            for id_2, entries_2 in instance2point_2.items():
                points2 = entries_2["points"]
                intersection = np.where(np.all(points2 >= min_bound, axis=1) & np.all(points2 <= max_bound, axis=1))[0].shape[0]
                if intersection > 0:
                    union = len(np.unique(np.concatenate((points1, points2))))            # flattened: distinct scalars
                    iou = float(intersection) / float(union)
                    if iou > 0.01:
                        id_pairs_iou.append((id_1, (id_2, iou)))
        ids_chunk_1, ids_chunk_2, ious = [], [], []
        for id1, (id2, iou) in id_pairs_iou:                                             # :465-477
            if id2 not in ids_chunk_2:
                ids_chunk_1.append(id1)
                ids_chunk_2.append(id2)
                ious.append(iou)
            else:
                i = ids_chunk_2.index(id2)
                if iou > ious[i]:
                    ious[i] = iou
                    ids_chunk_1[i] = id1
        for id1, id2 in zip(ids_chunk_1, ids_chunk_2):                                   # :479-481
            colors_2[instance2point_2[id2]["inds"]] = unique_colors_1[id1]
        merge_p = np.concatenate([merge_p, points_2])                                    # :488
        merge_c = np.concatenate([merge_c, colors_2])
        merge_p, merge_c = remove_duplicated_points(merge_p, merge_c)                    # :489
    return merge_p, merge_c
