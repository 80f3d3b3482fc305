#!/usr/bin/env python3
"""Asynchronous frontier against the level-synchronous driver on many random chunks (GPU box): identical labels expected.

    python tests/tools/fuzz_drivers.py [chunks] [seed]   -> one JSON line (profiles/r03_driver_fuzz.json)

Each driver runs in its own process (AI_NCUT_LOCKSTEP is read once per process); chunks: sizes logU(2k, 60k), the three
shipped configurations (TARL+Spatial T = 0.03, Spatial T = 0.075, tri-modal T = 0.005) in turn; the frontier runs them
both one by one and as batched calls of 8 with a small admission window."""
import json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MODES = [dict(name="tarl", tarl=True, dino=False, alpha=1.0, theta=0.5, gamma=0.0, T=0.03),
         dict(name="spatial", tarl=False, dino=False, alpha=1.0, theta=0.0, gamma=0.0, T=0.075),
         dict(name="tri", tarl=True, dino=True, alpha=1.0, theta=0.5, gamma=0.1, T=0.005)]


def worker(nchunks, seed, out, batched):
    sys.path.insert(0, ROOT)
    from autoinst_amd import ncuts_api as api, synth
    rng = np.random.default_rng(seed)
    sizes = [int(x) for x in np.exp(rng.uniform(np.log(2000), np.log(60000), nchunks))]
    res = {}
    graphs, metas = [], []
    for i, n in enumerate(sizes):
        m = MODES[i % 3]
        ch = synth.synthetic_chunk(n, seed=10_000 + seed * 1000 + i, tarl=m["tarl"], dino=m["dino"])
        g = api.build_affinity(ch["points"], ch["tarl"], ch["dino"], alpha=m["alpha"], theta=m["theta"], gamma=m["gamma"])
        if not batched:
            lab, ng, st = api.ncuts_labels(g, n, m["T"])
            res[f"c{i}"] = lab
            g.free()
        else:
            graphs.append(g); metas.append((i, n, m))
    if batched:
        for m in MODES:   # one T per batched call
            idx = [k for k, (i, n, mm) in enumerate(metas) if mm is m]
            for j in range(0, len(idx), 8):
                part = idx[j:j + 8]
                labs, ngs, st = api.ncuts_labels_batch([graphs[k] for k in part], None, m["T"], window_rows=60_000)
                for k, lab in zip(part, labs):
                    res[f"c{metas[k][0]}"] = lab
        for g in graphs:
            g.free()
    np.savez(out, **res)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5] == "1")
        sys.exit(0)
    nchunks = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    tmp = tempfile.mkdtemp()
    runs = {"lockstep": ("1", "0"), "frontier": ("0", "0"), "frontier_batched": ("0", "1")}
    out = {}
    for name, (lock, batched) in runs.items():
        path = os.path.join(tmp, name + ".npz")
        env = dict(os.environ, AI_NCUT_LOCKSTEP=lock)
        if lock == "1":   # the level-synchronous driver lives in the test-only build (make -C autoinst_amd/csrc lockstep)
            env["AUTOINST_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "autoinst_amd", "libautoinst_hip_lockstep.so")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(nchunks), str(seed), path, batched], check=True, env=env, timeout=1500)
        out[name] = np.load(path)
    ref = out["lockstep"]
    diff = {k: [c for c in ref.files if not np.array_equal(ref[c], out[k][c])] for k in ("frontier", "frontier_batched")}
    groups = int(sum(int(ref[c].max()) + 1 for c in ref.files))
    print(json.dumps({"chunks": nchunks, "seed": seed, "points": int(sum(ref[c].shape[0] for c in ref.files)), "groups": groups,
                      "modes": [m["name"] for m in MODES], "different_from_lockstep": diff,
                      "identical": all(len(v) == 0 for v in diff.values())}))
