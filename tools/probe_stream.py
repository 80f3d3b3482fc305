"""B prebuilt 200k-point TARL+Spatial graphs through ONE ai_ncut_batch call (admission window = AI_NCUT_WINDOW_ROWS):
python tools/probe_stream.py [reps] [B]  -> ms per call (graphs built outside the timed region), chunks/s of the cut alone"""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda", 0)
data = []
for k in range(min(B, 24)):
    ch = synth.synthetic_chunk(200_000, k, tarl=True)
    data.append((torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev)))
torch.cuda.synchronize()
ctx = api.Context(0)
graphs = [api.build_affinity(*data[k % len(data)], alpha=1.0, theta=0.5, gamma=0.0, ctx=ctx) for k in range(B)]
for rep in range(reps + 1):
    t0 = time.perf_counter()
    labs, ngs, st = api.ncuts_labels_batch(graphs, None, 0.03, window_rows=int(os.environ.get("AI_NCUT_WINDOW_ROWS", "0")) or None, time_spmv=bool(os.environ.get("AI_PROBE_CLOCK")))
    dt = time.perf_counter() - t0
    print(json.dumps({"ms": 1e3 * dt, "chunks_per_s_cut_only": B / dt, "window": os.environ.get("AI_NCUT_WINDOW_ROWS"),
                      **{k: st[k] for k in ("ms_total", "ms_eigen", "ms_spmv", "spmv_rows", "spmv_nnz", "levels", "lanczos_steps")}}), flush=True)
