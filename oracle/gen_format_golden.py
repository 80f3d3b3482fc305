"""Golden files for the on-disk formats next to the hot path, read back by the reference's readers.

TEST INFRASTRUCTURE.  Run in the build container only (needs /root/reference):

    python oracle/gen_format_golden.py

`pipeline/dataset/kitti_odometry_dataset.py` cannot be imported (pykitti, nptyping: ordinary ModuleNotFoundError), but its
reader methods `get_tarl_features` (:251-281) and `get_dinov2_features` (:224-249) need os, zlib and numpy only.  The two
`def` nodes are compiled from the reference file where it lies (found with `ast`; nothing else of the module runs) and
called with a stand-in `self` that holds the two directory attributes they read.  The files they read are written by
`autoinst_amd.formats`; what is stored (tests/golden/formats/) is those small files plus the arrays the REFERENCE readers
returned from them.  tests/test_formats.py then checks that our readers return the same arrays from the same files, and
that our writers reproduce the files byte for byte where the container format is deterministic (the zlib .bin).
"""
from __future__ import annotations

import ast
import os
import sys
import types
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoinst_amd import formats  # noqa: E402

REF_FILE = "/root/reference/pipeline/dataset/kitti_odometry_dataset.py"
OUT = os.path.join(ROOT, "tests", "golden", "formats")


def reference_methods(path, names):
    found = {}
    for node in ast.walk(ast.parse(open(path).read())):
        if isinstance(node, ast.FunctionDef) and node.name in names:
            env = {"np": np, "os": os, "zlib": zlib}
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), env)
            found[node.name] = env[node.name]
    missing = set(names) - set(found)
    if missing:
        raise SystemExit(f"not found in {path}: {missing}")
    return found


def main():
    ref = reference_methods(REF_FILE, ("get_tarl_features", "get_dinov2_features"))
    os.makedirs(os.path.join(OUT, "tarl"), exist_ok=True)
    os.makedirs(os.path.join(OUT, "dino", "image_2"), exist_ok=True)
    rng = np.random.default_rng(77)
    me = types.SimpleNamespace(tarl_features_path=os.path.join(OUT, "tarl"), dinov2_features_path=os.path.join(OUT, "dino"))

    feats = rng.standard_normal((211, 96)).astype(np.float32)
    feats[::17] = 0.0                                     # points the extractor never reached
    formats.write_tarl_bin(os.path.join(OUT, "tarl", "000042.bin"), feats)
    got = ref["get_tarl_features"](me, 42)
    assert got.dtype == np.float32 and np.array_equal(got, feats)

    fmap = rng.standard_normal((7, 11, 384)).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "dino", "image_2", "000007.npz"), feature_map=fmap)
    got_d = ref["get_dinov2_features"](me, "cam2", 7)
    assert np.array_equal(got_d, fmap)

    np.savez_compressed(os.path.join(OUT, "expected.npz"), tarl_000042=got, dino_cam2_000007=got_d)
    print("tarl", got.shape, "dino", got_d.shape, "-> the reference readers return what was written")


if __name__ == "__main__":
    sys.exit(main())
