"""Golden vectors for the SAM factor, produced by the reference's own function.

TEST INFRASTRUCTURE.  Run in the build container only (needs /root/reference):

    python oracle/gen_sam_golden.py

`pipeline/utils/image/image_utils.py` cannot be imported here as a module (its first line imports cv2, then open3d:
ordinary ModuleNotFoundError), but `sam_label_distance` (:64-89) itself needs numpy only.  The function object is
therefore compiled from the reference file where it lies -- the one `def` node, found with `ast`, nothing else of the
module executed -- and run on seeded inputs; only the inputs and its outputs are stored (tests/golden/sam_ref.npz).
Nothing of the reference's text enters the repository.
"""
from __future__ import annotations

import ast
import os
import sys

import numpy as np
from scipy.spatial.distance import cdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_FILE = "/root/reference/pipeline/utils/image/image_utils.py"


def reference_function(path, name):
    src = open(path).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            env = {"np": np}
            exec(compile(mod, path, "exec"), env)
            return env[name]
    raise SystemExit(f"{name} not found in {path}")


def main():
    ref = reference_function(REF_FILE, "sam_label_distance")
    out = {}
    for case, (n, views, seed, beta, radius, spread) in enumerate(
            [(300, 5, 21, 0.7, 1.0, 1.5), (220, 1, 22, 2.0, 1.0, 1.0), (260, 8, 23, 1.3, 0.6, 1.2)]):
        rng = np.random.default_rng(seed)
        pts = rng.normal(0, spread, (n, 3))
        pts[5] = pts[4]                                   # duplicate point
        pts[7] = pts[6] + np.array([radius, 0.0, 0.0])    # on the radius (<=, up to cdist's rounding)
        ids = rng.integers(0, 4, (n, views))
        ids[rng.random((n, views)) < 0.35] = -1           # not seen in that view
        ids[11] = -1                                      # never seen
        sd = cdist(pts, pts)
        label_distance, mask = ref(ids, sd, radius, beta)
        r, c = np.nonzero(mask)
        out.update({f"c{case}_points": pts, f"c{case}_sam": ids.astype(np.int64), f"c{case}_beta": beta,
                    f"c{case}_radius": radius, f"c{case}_rows": r.astype(np.int32), f"c{case}_cols": c.astype(np.int32),
                    f"c{case}_factor": label_distance[r, c]})
        assert (label_distance[mask == 0] == 0).all()
        print(f"case {case}: n {n} views {views} pairs {r.size} factor<1 on {(label_distance[r, c] < 1).sum()}")
    out["cases"] = 3
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sam_ref.npz"), **out)


if __name__ == "__main__":
    sys.exit(main())
