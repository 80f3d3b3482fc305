"""Cases of the GPU suite that need the fault-injection hooks of the TEST-ONLY build (libautoinst_hip_lockstep.so, compiled with
-DAI_TEST_HOOKS): run as a child process by tests/test_gpu_parity.py with AUTOINST_HIP_LIB pointing at that build.

    python tests/hook_cases.py slots | fences
"""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoinst_amd import ncuts_api as api, synth  # noqa: E402


class _Env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        for k, v in self.kv.items():
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k in self.kv:
            os.environ.pop(k, None)


def _capture_stderr(fn):
    """fn() with fd 2 redirected to a file (the library writes with fprintf)."""
    import tempfile
    sys.stderr.flush()
    with tempfile.TemporaryFile(mode="w+b") as f:
        saved = os.dup(2)
        os.dup2(f.fileno(), 2)
        try:
            out = fn()
        finally:
            os.dup2(saved, 2)
            os.close(saved)
        f.seek(0)
        return out, f.read().decode("utf-8", "replace")


def slots():
    rng = np.random.default_rng(11)
    sizes = [int(x) for x in np.exp(rng.uniform(np.log(1500), np.log(5000), 600))][:40]
    chunks = [synth.synthetic_chunk(n, 500 + i, tarl=False) for i, n in enumerate(sizes)]
    graphs = [api.build_affinity(c["points"], None, alpha=1.0, theta=0.0, gamma=0.0) for c in chunks]
    labs, ngs, st = api.ncuts_labels_batch(graphs, None, 0.075)
    assert st["unconverged"] == 0
    # 48 slots: labels unchanged, children waited
    with _Env(AI_FLOW_SMAX=48, AI_NCUT_PHASES=1):
        (labs48, ngs48, st48), err = _capture_stderr(lambda: api.ncuts_labels_batch(graphs, None, 0.075))
    waited = int(re.search(r"children that waited for a slot (\d+)", err).group(1))
    assert waited > 0, err
    assert ngs48 == ngs and all(np.array_equal(a, b) for a, b in zip(labs48, labs))
    # num_points_orig = 1 % of the size: segments down to 0.01 % of the chunk stay eligible
    big = graphs[:8]
    norig = [max(1, g.n // 100) for g in big]
    labs_s, ngs_s, st_s = api.ncuts_labels_batch(big, norig, 0.075)
    with _Env(AI_FLOW_SMAX=32):
        labs_t, ngs_t, _ = api.ncuts_labels_batch(big, norig, 0.075)
    assert ngs_t == ngs_s and all(np.array_equal(a, b) for a, b in zip(labs_t, labs_s))
    for g in graphs:
        g.free()


def fences():
    """AI_FLOW_INJECT=k spoils the Ritz coefficients of the k-th harvested segment of a call, the way a check that froze a segment on bad
    data would: the segment is solved again (same graph, same start vector), the labels are those of the undisturbed call, and
    stats['restarted_solves'] says it happened."""
    ch = synth.synthetic_chunk(30000, 21, tarl=True)
    g = api.build_affinity(ch["points"], ch["tarl"], alpha=1.0, theta=0.5, gamma=0.0)
    lab0, ng0, st0 = api.ncuts_labels(g, g.n, 0.03)
    assert st0["restarted_solves"] == 0 and st0["unconverged"] == 0
    assert st0["lanczos_solves"] > 12
    for k in (0, 1, 5, 12):
        with _Env(AI_FLOW_INJECT=k):
            lab, ng, st = api.ncuts_labels(g, g.n, 0.03)
        assert st["restarted_solves"] == 1, (k, st)
        assert ng == ng0 and np.array_equal(lab, lab0), k
        assert st["lanczos_solves"] == st0["lanczos_solves"] and st["unconverged"] == 0
        assert st["max_true_resid"] <= 2e-10   # of the pairs that were CUT: the spoiled one was not
    # a limit below every residual sends EVERY segment back once; the repeat must reproduce the residual bit for bit to be accepted
    # (three different residuals end the call with an error): every solve of the chunk is shown to be reproducible inside one call
    with _Env(AI_FLOW_TRUE_LIMIT="1e-300"):
        lab, ng, st = api.ncuts_labels(g, g.n, 0.03)
    assert ng == ng0 and np.array_equal(lab, lab0)
    assert st["restarted_solves"] >= 0.9 * st0["lanczos_solves"] - 2 and st["lanczos_solves"] == st0["lanczos_solves"], (st, st0)
    assert st["accepted_above_limit"] >= 0.9 * st0["lanczos_solves"] - 2 and st["true_resid_limit"] == 1e-300, st
    # the packed histories of a wave carry the device's size of T and an integer checksum per row; AI_FLOW_INJECT_HIST=k makes the host
    # reject the rows of the wave that holds the k-th harvested segment once: they are packed again, nothing else changes
    for k in (0, 7):
        with _Env(AI_FLOW_INJECT_HIST=k):
            lab, ng, st = api.ncuts_labels(g, g.n, 0.03)
        assert st["hist_retries"] == 1 and st["restarted_solves"] == 0, (k, st)
        assert ng == ng0 and np.array_equal(lab, lab0), k
    assert st0["hist_retries"] == 0
    # in a batched call too (the spoiled segment belongs to one of several chunks)
    chunks = [synth.synthetic_chunk(n, 40 + i, tarl=False) for i, n in enumerate((9000, 14000, 5000))]
    gs = [api.build_affinity(c["points"], None, alpha=1.0, theta=0.0, gamma=0.0) for c in chunks]
    labs0, ngs0, stb0 = api.ncuts_labels_batch(gs, None, 0.05)
    with _Env(AI_FLOW_INJECT=4):
        labs1, ngs1, stb1 = api.ncuts_labels_batch(gs, None, 0.05)
    assert stb0["restarted_solves"] == 0 and stb1["restarted_solves"] == 1
    assert ngs1 == ngs0 and all(np.array_equal(a, b) for a, b in zip(labs1, labs0))
    for x in gs + [g]:
        x.free()


if __name__ == "__main__":
    case = sys.argv[1]
    {"slots": slots, "fences": fences}[case]()
    print(f"hook case {case}: ok")
