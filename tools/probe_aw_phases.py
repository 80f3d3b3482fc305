"""Where a block of k_weights_lanes spends its time (device clock, wave 0 of every block), from a build with -DAW_PHASES:
    make -C autoinst_amd/csrc phases && AUTOINST_HIP_LIB=autoinst_amd/libautoinst_hip_awphases.so python tools/probe_aw_phases.py
"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth, _ffi
lib = _ffi.load()
dev = torch.device("cuda", 0)
ctx = api.Context(0)
ch = synth.synthetic_chunk(200_000, seed=0, tarl=True, dino=True)
P, T, D = (torch.from_numpy(ch[k]).to(dev) for k in ("points", "tarl", "dino"))
names = ["hash build", "numbering + slot->position", "row set-up + offsets", "slabs (stage + distances)", "8-lane reductions", "weights (exp, store)"]
buf = (C.c_ulonglong * 8)()
for name, kw in (("cfg2 tarl 96", dict(tarl=T, dino=None, theta=0.5, gamma=0.0)), ("cfg4 tarl 96 + dino 384", dict(tarl=T, dino=D, theta=0.5, gamma=0.1))):
    g = api.build_affinity(P, kw["tarl"], kw["dino"], alpha=1.0, theta=kw["theta"], gamma=kw["gamma"], ctx=ctx); g.free()
    torch.cuda.synchronize(); lib.ai_debug_aw_phases(buf, 1)
    reps = 5
    for _ in range(reps):
        g = api.build_affinity(P, kw["tarl"], kw["dino"], alpha=1.0, theta=kw["theta"], gamma=kw["gamma"], ctx=ctx); g.free()
    torch.cuda.synchronize(); lib.ai_debug_aw_phases(buf, 1)
    nblk = (200_000 + 15) // 16
    tot = sum(buf[:6])
    print(name, "per block, 100 MHz ticks x 10 ns:", "; ".join(f"{n} {buf[i] / reps / nblk * 10:.0f} ns ({100 * buf[i] / tot:.0f} %)" for i, n in enumerate(names)), f"; block total {tot / reps / nblk * 10:.0f} ns")
