"""One-off full-size parity run (GPU box): HIP path vs the CPU oracle on the 200k-point chunk.

    python tools/fullsize_parity.py [N] [tarl|spatial]  -> gpurun_out/fullsize_parity_<N>_<mode>.json
The oracle (scipy eigsh shift-invert, as the reference) needs many minutes at 200k; progress goes
to stdout so that the run is not taken for hung.
"""
import json, sys, time, threading
import numpy as np
sys.path.insert(0, ".")
from autoinst_amd import ncuts_api as api, synth
from oracle import ncuts_ref, metrics_ref

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
mode = sys.argv[2] if len(sys.argv) > 2 else "tarl"
theta, T = (0.5, 0.03) if mode == "tarl" else (0.0, 0.075)
ch = synth.synthetic_chunk(n, 0, tarl=(mode == "tarl"))
gt = ch["gt"].copy(); gt[::97] = 0
t0 = time.perf_counter()
groups = api.ncuts(ch["points"], ch["tarl"], alpha=1.0, theta=theta, gamma=0.0, T=T)
t_gpu = time.perf_counter() - t0
lab = ncuts_ref.groups_to_labels(groups, n)
print("gpu done", t_gpu, len(groups), flush=True)
stop = False
def ticker():
    while not stop:
        print("oracle running", round(time.perf_counter() - t0), "s", flush=True); time.sleep(30)
threading.Thread(target=ticker, daemon=True).start()
t1 = time.perf_counter()
A = ncuts_ref.affinity_sparse(ch["points"], ch["tarl"], alpha=1.0, theta=theta, gamma=0.0)
t_aff = time.perf_counter() - t1
print("oracle affinity", t_aff, flush=True)
st = {}
ref = ncuts_ref.normalized_cut(A, n, np.arange(n), T=T, fast=True, stats=st)
t_cpu = time.perf_counter() - t1
stop = True
rl = ncuts_ref.groups_to_labels(ref, n)
cg, cr = ncuts_ref.canonical_labels(lab) + 1, ncuts_ref.canonical_labels(rl) + 1
out = {"n": n, "mode": mode, "gpu_seconds_incl_upload": t_gpu, "cpu_oracle_seconds": t_cpu, "cpu_affinity_seconds": t_aff,
       "cpu_eigsh_calls": st.get("eigsh"), "groups_gpu": len(groups), "groups_cpu": len(ref),
       "partition_equal": bool(ncuts_ref.partitions_equal(lab, rl)), "ARI": ncuts_ref.adjusted_rand_index(lab, rl),
       "scores_gpu": metrics_ref.score(cg, cg, gt), "scores_cpu": metrics_ref.score(cr, cr, gt)}
print(json.dumps(out), flush=True)
open(f"gpurun_out/fullsize_parity_{n}_{mode}.json", "w").write(json.dumps(out, indent=1))
