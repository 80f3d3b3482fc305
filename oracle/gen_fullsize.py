"""ORACLE tooling -- full-size goldens: the UNMODIFIED oracle recursion at 50k / 100k / 200k points.

    python oracle/gen_fullsize.py N MODE SEED [--prewarm K] [--perm P] [--tag TAG]
        -> tests/golden/full_<N>_<MODE>_<SEED>[_<TAG>].npz

MODE is ``tarl`` (alpha=1, theta=0.5, T=0.03: config_tarl_spatial, reference ``pipeline/config.py:17-26``)
``spatial`` (alpha=1, T=0.075: config_spatial, ``config.py:28-37``) or ``tri`` (plus gamma=0.1 on 384-d DINO
features, T=0.005: config_tarl_spatial_dino, ``config.py:6-15``).  The chunk comes from the seeded
generator (``autoinst_amd/synth.py``), the affinity from ``ncuts_ref.affinity_sparse``
(``ncuts_utils.py:60-67,135-156,167``) and the partition from ``ncuts_ref.normalized_cut``
(``normalized_cut.py:37-63``: SciPy ``eigsh(A, 2, sigma=1e-10, which='LM')`` on EVERY segment, connected
or not, exactly as the reference; the only restated part is ``D.todense()[cut].sum()`` -> ``d[cut].sum()``,
pinned on the small fixtures by ``gen_golden.py``).  The reference module itself cannot run at these
sizes: ``normalized_cut.py:9-10`` densifies an N x N diagonal matrix (320 GB at 200k).

This is a CPU-only job for the build container: one fresh interpreter per fixture (ARPACK keeps its
start-vector RNG state between ``eigsh`` calls of a process, and on a disconnected segment the returned
null-space vector depends on it).  ``--prewarm K`` runs K throw-away ``eigsh`` calls first, i.e. puts
ARPACK into another state: the spread of the oracle's labels over K shows how far the reference
disagrees with ITSELF on the same chunk (stored under a different ``--tag``).  ``--perm P`` lists the SAME points
in another order (seeded permutation P; labels are stored in the original order): the chunk is physically the
same, only SuperLU's elimination order and round-off change.

Stored: labels (canonical, int32), group count, scores against the synthetic ground truth, wall
seconds per stage, eigsh calls, core count / CPU model / library versions.  Only data is committed.
"""
from __future__ import annotations

import json
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

import numpy as np  # noqa: E402
import scipy  # noqa: E402
import scipy.sparse as sp  # noqa: E402
import scipy.sparse.linalg as spla  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from autoinst_amd import synth  # noqa: E402
from oracle import metrics_ref, ncuts_ref  # noqa: E402

MODES = {"tarl": dict(alpha=1.0, theta=0.5, gamma=0.0, T=0.03),
         "spatial": dict(alpha=1.0, theta=0.0, gamma=0.0, T=0.075),
         "tri": dict(alpha=1.0, theta=0.5, gamma=0.1, T=0.005)}   # config_tarl_spatial_dino, config.py:6-15


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def chunk_for(n, mode, seed):
    """The seeded synthetic chunk of a fixture (features only where the mode uses them)."""
    return synth.synthetic_chunk(n, seed, tarl=(mode in ("tarl", "tri")), dino=(mode == "tri"))


def scoring_gt(gt):
    """Ground truth used by every full-size comparison: every 97th point is background."""
    gt = np.array(gt, copy=True)
    gt[::97] = 0
    return gt


def main(argv):
    n, mode, seed = int(argv[0]), argv[1], int(argv[2])
    prewarm, tag, perm = 0, "", None
    rest = argv[3:]
    while rest:
        if rest[0] == "--prewarm":
            prewarm = int(rest[1])
        elif rest[0] == "--tag":
            tag = rest[1]
        elif rest[0] == "--perm":
            perm = int(rest[1])
        rest = rest[2:]
    cfg = MODES[mode]
    name = f"full_{n}_{mode}_{seed}" + (f"_{tag}" if tag else "")
    out = os.path.join(ROOT, "tests", "golden", name + ".npz")
    log = lambda *a: print(name, *a, flush=True)

    if prewarm:
        rng = np.random.default_rng(12345)
        for _ in range(prewarm):
            M = sp.random(200, 200, density=0.1, random_state=rng, format="csr")
            M = M + M.T + sp.identity(200) * 50
            spla.eigsh(M, 2, sigma=1e-10, which="LM")

    ch = chunk_for(n, mode, seed)
    gt_orig = ch["gt"]
    order = None
    if perm is not None:
        order = np.random.default_rng(perm).permutation(n)
        ch = {k: (None if v is None else v[order]) for k, v in ch.items()}
    t0 = time.perf_counter()
    A = ncuts_ref.affinity_sparse(ch["points"], ch["tarl"], ch["dino"], alpha=cfg["alpha"], theta=cfg["theta"], gamma=cfg["gamma"])
    t_aff = time.perf_counter() - t0
    log("affinity", round(t_aff, 1), "s nnz", A.nnz)
    st = {}
    t1 = time.perf_counter()
    groups = ncuts_ref.normalized_cut(A, n, np.arange(n), T=cfg["T"], split_lim=0.01, fast=True, stats=st)
    t_cut = time.perf_counter() - t1
    lab = ncuts_ref.groups_to_labels(groups, n)
    if order is not None:   # back to the generator's point order
        back = np.empty(n, dtype=np.int64)
        back[order] = lab
        lab = back
    lab = ncuts_ref.canonical_labels(lab).astype(np.int32)
    sc = metrics_ref.score(lab + 1, lab + 1, scoring_gt(gt_orig))
    meta = {"n": n, "mode": mode, "seed": seed, "cfg": cfg, "prewarm": prewarm, "perm": perm, "nnz": int(A.nnz),
            "groups": len(groups), "eigsh_calls": st.get("eigsh", 0),
            "affinity_seconds": t_aff, "normalized_cut_seconds": t_cut,
            "cores_used": 1, "nproc": os.cpu_count(), "cpu_model": cpu_model(),
            "concurrent_jobs": int(os.environ.get("AI_CONCURRENT_JOBS", "1")),
            "numpy": np.__version__, "scipy": scipy.__version__, "scores": sc}
    log(json.dumps(meta))
    np.savez_compressed(out, labels=lab, meta=json.dumps(meta))


if __name__ == "__main__":
    main(sys.argv[1:])
