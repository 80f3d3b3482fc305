"""On-disk formats next to the hot path: round trips and layout checks."""
import zlib

import numpy as np

from autoinst_amd import formats


def test_tarl_bin_is_zlib_float32(tmp_path):
    f = np.random.default_rng(0).standard_normal((57, 96)).astype(np.float32)
    p = tmp_path / "000003.bin"
    formats.write_tarl_bin(p, f)
    # exactly what tarl_extractor.py:84-89 writes and kitti_odometry_dataset.py:251-281 reads
    raw = np.frombuffer(zlib.decompress(p.read_bytes()), dtype=np.float32).reshape(-1, 96)
    assert np.array_equal(raw, f) and np.array_equal(formats.read_tarl_bin(p), f)


def test_selftrain_npz_keys(tmp_path):
    p = tmp_path / "s.npz"
    formats.write_selftrain_npz(p, np.zeros((5, 3)), np.arange(5), np.arange(5) * 2, np.ones(5))
    z = formats.read_selftrain_npz(p)
    assert set(z) == {"pts", "ncut_labels", "kitti_labels", "cluster_labels", "semantic"}
    assert np.array_equal(z["cluster_labels"], np.zeros(5, dtype=z["ncut_labels"].dtype))


def test_pcd_round_trip_and_label_colors(tmp_path):
    rng = np.random.default_rng(1)
    pts = rng.standard_normal((100, 3))
    lab = rng.integers(0, 7, 100)
    col = formats.labels_to_colors(lab)
    assert np.all(col[lab == 0] == 0)
    p = tmp_path / "c.pcd"
    formats.write_pcd_binary(p, pts, col)
    head = p.read_bytes()[:200].decode("ascii", "replace")
    assert "FIELDS x y z rgb" in head and "DATA binary" in head
    p2, c2 = formats.read_pcd_binary(p)
    assert np.allclose(p2, pts.astype(np.float32)) and np.array_equal(c2, col)
    # colours identify the instance again (run_pipeline.py:214-217 recovers labels with np.unique over colours)
    _, back = np.unique(c2, axis=0, return_inverse=True)
    assert len(np.unique(back)) == len(np.unique(lab))
    for i in np.unique(lab):
        assert len(np.unique(back[lab == i])) == 1


def test_dino_npz_key(tmp_path):
    p = tmp_path / "000001.npz"
    fm = np.zeros((4, 5, 384), dtype=np.float32)
    np.savez(p, feature_map=fm)
    assert formats.read_dino_npz(p).shape == (4, 5, 384)


def test_files_as_the_reference_readers_see_them(tmp_path):
    """tests/golden/formats/: small files + the arrays the REFERENCE's own readers (kitti_odometry_dataset.py:251-281,
    :224-249, compiled from the reference file by oracle/gen_format_golden.py) returned from them.  Our readers return the
    same arrays; our TARL writer reproduces the file byte for byte."""
    import os
    from conftest import GOLDEN
    d = os.path.join(GOLDEN, "formats")
    exp = np.load(os.path.join(d, "expected.npz"))
    t = formats.read_tarl_bin(os.path.join(d, "tarl", "000042.bin"))
    assert t.dtype == np.float32 and np.array_equal(t, exp["tarl_000042"])
    assert np.array_equal(formats.read_dino_npz(os.path.join(d, "dino", "image_2", "000007.npz")), exp["dino_cam2_000007"])
    p = tmp_path / "000042.bin"
    formats.write_tarl_bin(p, exp["tarl_000042"])
    assert p.read_bytes() == open(os.path.join(d, "tarl", "000042.bin"), "rb").read()
