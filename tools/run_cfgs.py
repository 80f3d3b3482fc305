#!/usr/bin/env python3
"""BASELINE.json configs[0], [3] on one GPU (configs[1] is bench.py, configs[2] tools/run_cfg3.py, configs[4] tests/tools/run_cfg5.py).

    python tools/run_cfgs.py            # prints one JSON line per config
"""
import json, os, sys, time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth


def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return r, (time.perf_counter() - t0) / reps


def main():
    import torch
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)   # torch first: it will not initialise HIP after the library has
    api.default_context()
    # cfg1: 10k points, spatial only, T = 0.075
    ch = synth.synthetic_chunk(10_000, 0, tarl=False)
    (groups, dt) = timed(lambda: api.ncuts(ch["points"], None, alpha=1.0, theta=0.0, gamma=0.0, T=0.075))
    print(json.dumps({"config": "cfg1 10k spatial", "ms": 1e3 * dt, "groups": len(groups)}), flush=True)
    # cfg4: 200k points, TARL + spatial + DINO(384-d), T = 0.005
    ch = synth.synthetic_chunk(200_000, 0, tarl=True, dino=True)

    def cfg4():
        t0 = time.perf_counter()
        g = api.build_affinity(ch["points"], ch["tarl"], ch["dino"], alpha=1.0, theta=0.5, gamma=0.1)
        t1 = time.perf_counter()
        lab, ng, st = api.ncuts_labels(g, 200_000, 0.005)
        g.free()
        return {"affinity_ms": 1e3 * (t1 - t0), "ncut_ms": st["ms_total"], "groups": ng, "steps": st["lanczos_steps"], "nnz": None}
    r, dt = timed(cfg4)
    print(json.dumps({"config": "cfg4 200k TARL+spatial+DINO384 (host arrays: 770 MB of features uploaded per call)", "ms": 1e3 * dt, **r}), flush=True)
    tp, tt, td = (torch.from_numpy(ch[k]).to(dev) for k in ("points", "tarl", "dino"))

    def cfg4_dev():
        g = api.build_affinity(tp, tt, td, alpha=1.0, theta=0.5, gamma=0.1)
        g.free()
    _, dt = timed(cfg4_dev)
    print(json.dumps({"config": "cfg4 affinity only, features resident in HBM", "ms": 1e3 * dt}), flush=True)

    def cfg4_res():
        t0 = time.perf_counter()
        g = api.build_affinity(tp, tt, td, alpha=1.0, theta=0.5, gamma=0.1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        lab, ng, st = api.ncuts_labels(g, 200_000, 0.005)
        g.free()
        return {"affinity_ms": 1e3 * (t1 - t0), "ncut_ms": st["ms_total"], "groups": ng, "steps": st["lanczos_steps"]}
    r, dt = timed(cfg4_res)
    print(json.dumps({"config": "cfg4 200k TARL+spatial+DINO384, inputs resident in HBM", "ms": 1e3 * dt, **r}), flush=True)
    del tp, tt, td, ch
    # cfg5 (1M points, k = 64) has its own driver: tests/tools/run_cfg5.py


if __name__ == "__main__":
    main()
