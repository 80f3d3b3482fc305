import sys, time, json
sys.path.insert(0, ".")
from autoinst_amd import ncuts_api as api, synth
api.default_context()
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    ch = synth.synthetic_chunk(200000, seed, tarl=True)
    g = api.build_affinity(ch["points"], ch["tarl"], alpha=1.0, theta=0.5, gamma=0.0)
    api.ncuts_labels(g, 200000, 0.03)
    t0 = time.perf_counter(); lab, ng, st = api.ncuts_labels(g, 200000, 0.03); dt = time.perf_counter() - t0
    print(json.dumps({"seed": seed, "ms": dt * 1e3, "steps": st["lanczos_steps"], "levels": st["levels"], "groups": ng, "unconv": st["unconverged"], "maxres": st["max_resid"]}), flush=True)
    g.free()
