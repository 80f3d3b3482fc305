"""Opt-in soak (-m gpu with AI_SOAK=1; AI_SOAK_SECONDS, default 180): the same batched calls again and again from two host threads,
every repeat compared with the first -- labels, group counts, the device's work counters and the largest accepted residual.
Round 3's race (a convergence check that read its pool entry through the scalar cache: one chunk in ~700 came out with other
labels) was invisible to the 25-repeat test of the GPU suite and showed within a minute of this; `tools/soak_trace.py` is the
variant that names the segments that differ."""
import os
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(os.environ.get("AI_SOAK") != "1", reason="opt-in: AI_SOAK=1 (about AI_SOAK_SECONDS = 180 s of GPU time)")
def test_soak_two_threads_thin_tail_chunks():
    from autoinst_amd import ncuts_api as api, synth
    budget = float(os.environ.get("AI_SOAK_SECONDS", "180"))
    # thin-tail chunks (the regime that raced: relists follow each other within microseconds at the end of a call) and one large one
    sets = [[(30000, 1), (24000, 2), (30000, 3), (18000, 4)], [(60000, 5), (9000, 6), (12000, 7), (200000, 0)]]
    results = [None, None]
    errors = []

    def run(w):
        try:
            ctx = api.Context(0)
            chunks = [synth.synthetic_chunk(n, seed, tarl=True) for n, seed in sets[w]]
            first, reps, t0 = None, 0, time.time()
            while time.time() - t0 < budget:
                graphs = [api.build_affinity(c["points"], c["tarl"], alpha=1.0, theta=0.5, gamma=0.0, ctx=ctx) for c in chunks]
                labs, ngs, st = api.ncuts_labels_batch(graphs, None, 0.03)
                for g in graphs:
                    g.free()
                cur = (labs, ngs, st["spmv_rows"], st["spmv_nnz"], st["lanczos_solves"], st["max_resid"], st["unconverged"])
                if first is None:
                    first = cur
                else:
                    lab_diff = [int((np.asarray(a) != np.asarray(b)).sum()) for a, b in zip(cur[0], first[0])]
                    assert cur[2:] == first[2:] and cur[1] == first[1] and not any(lab_diff), (
                        f"thread {w}, repeat {reps}: counters {cur[2:]} vs first {first[2:]}; groups {cur[1]} vs {first[1]}; "
                        f"labels that differ per chunk {lab_diff}")
                reps += 1
            results[w] = reps
            ctx.close()
        except BaseException as e:  # noqa: BLE001 -- reported by the test's own thread
            errors.append(e)

    ts = [threading.Thread(target=run, args=(w,)) for w in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errors:
        # keep the evidence where a GPU-box run merges it back (gpurun_out/), whatever the test runner prints
        try:
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
            with open(os.path.join(root, "gpurun_out", "soak_failure.txt"), "a") as f:
                for e in errors:
                    f.write(f"{time.strftime('%Y-%m-%d %H:%M:%S')} {type(e).__name__}: {e}\n")
        except OSError:
            pass
        raise errors[0]
    assert all(r and r >= 3 for r in results), results
    print(f"soak: {results} repeats of 4-chunk calls in {budget:.0f} s, every one equal to its first")
