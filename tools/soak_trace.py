"""Repeat one call with the per-segment trace on (AI_NCUT_DEBUG=2) and show the segments whose line differs from the first call's.
python tools/soak_trace.py R B first_seed [n_points] [tarl|spatial|tri]"""
import json, os, sys, tempfile
os.environ["AI_NCUT_DEBUG"] = "2"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoinst_amd import ncuts_api as api, synth
from bench import CFG, N_POINTS
R = int(sys.argv[1]); B = int(sys.argv[2]); c0 = int(sys.argv[3])
npts = int(sys.argv[4]) if len(sys.argv) > 4 else N_POINTS
MODES = {"tarl": dict(tarl=True, dino=False, alpha=1.0, theta=0.5, gamma=0.0, T=0.03),
         "spatial": dict(tarl=False, dino=False, alpha=1.0, theta=0.0, gamma=0.0, T=0.075),
         "tri": dict(tarl=True, dino=True, alpha=1.0, theta=0.5, gamma=0.1, T=0.005)}
M = MODES[sys.argv[5] if len(sys.argv) > 5 else "tarl"]      # the three shipped configurations
CFG = dict(CFG, alpha=M["alpha"], theta=M["theta"], gamma=M["gamma"], T=M["T"])
dev = torch.device("cuda", 0)
ctx = api.Context(0)
data = []
for c in range(c0, c0 + B):
    ch = synth.synthetic_chunk(npts, seed=c, tarl=M["tarl"], dino=M["dino"])
    data.append((torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev) if M["tarl"] else None,
                 torch.from_numpy(ch["dino"]).to(dev) if M["dino"] else None))
tmp = tempfile.TemporaryFile(mode="w+b")
saved = os.dup(2)
first, events = None, 0
for r in range(R):
    graphs = [api.build_affinity(p, f, d, alpha=CFG["alpha"], theta=CFG["theta"], gamma=CFG["gamma"], ctx=ctx) for p, f, d in data]
    tmp.seek(0); tmp.truncate()
    sys.stderr.flush()
    os.dup2(tmp.fileno(), 2)
    try:
        labs, ngs, st = api.ncuts_labels_batch(graphs, None, CFG["T"])
    finally:
        os.dup2(saved, 2)
        for g in graphs: g.free()
    tmp.seek(0)
    raw = tmp.read().decode().splitlines()
    for l in raw:
        if l.startswith("[anom]"):
            print(r, l, flush=True)
    lines = sorted(l for l in raw if l.startswith("[seg]"))
    if r % 300 == 299:
        print(f"progress: {r + 1} repeats, {events} events", flush=True)   # (a GPU-box run that stays silent for minutes is taken to be hung)
    if first is None:
        first = (lines, [np.asarray(l).copy() for l in labs])
        print("segments:", len(lines), flush=True)
    elif lines != first[0]:
        events += 1
        a, b = set(first[0]), set(lines)
        print(json.dumps({"r": r, "labels_differ": [int((x != np.asarray(y)).sum()) for x, y in zip(first[1], labs)],
                          "only_first": sorted(a - b)[:6], "only_now": sorted(b - a)[:6], "n_first": len(a - b), "n_now": len(b - a)}), flush=True)
        if events >= 6:
            break
print(json.dumps({"R": r + 1, "events": events}))
