#!/usr/bin/env python3
"""BASELINE.json configs[2]: a whole map's chunks, chunk-parallel across the GPUs of one node.

The sample map is not available offline, so the map is synthetic: 64 chunks with
the real size distribution N ~ logU(3k, 30k) plus 8 chunks of 200k points, TARL+Spatial.
Launch with one rank per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/run_cfg3.py
    python tools/run_cfg3.py --small            # 1 GPU, reduced map (used by the GPU test suite)

Chunks are assigned by LPT (autoinst_amd.sharding), each rank runs `--in-flight` host threads that
push `--batch` chunks per batched call, label arrays are gathered to rank 0 (RCCL), which prints
one JSON line.
"""
import argparse, json, os, sys, time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def chunk_sizes(small: bool):
    rng = np.random.default_rng(3)
    if small:
        return [int(x) for x in np.exp(rng.uniform(np.log(3000), np.log(12000), 6))] + [40_000]
    return [int(x) for x in np.exp(rng.uniform(np.log(3000), np.log(30000), 64))] + [200_000] * 8


def run_map(sizes, world, rank, local_rank, in_flight=2, dist=None, dev=None, batch=12, repeat=1):
    """-> (labels of the last pass, seconds of the first pass, seconds of the last pass).  The first pass of a process pays for
    the contexts' workspaces (collected block by block, then consolidated) and the code objects; a map is usually not the first."""
    from autoinst_amd import ncuts_api as api, sharding, synth
    mine = sharding.lpt_assign(sizes, world)[rank]
    ctxs = [api.Context(local_rank) for _ in range(in_flight)]
    data = [synth.synthetic_chunk(sizes[i], seed=1000 + i, tarl=True) for i in mine]
    times = []
    for _ in range(max(1, repeat)):
        t0 = time.perf_counter()
        # the rank's chunk loop: host threads take batches (largest chunks first) from one queue, one batched call per batch
        labs = sharding.run_chunks([(d["points"], d["tarl"]) for d in data], threads=in_flight, batch=batch, contexts=ctxs,
                                   alpha=1.0, theta=0.5, gamma=0.0, T=0.03)
        merged = sharding.gather_labels(dict(zip(mine, labs)), device=dev)
        times.append(time.perf_counter() - t0)
    return merged, times[0], times[-1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--in-flight", type=int, default=2, help="host threads (contexts) per GPU")
    ap.add_argument("--batch", type=int, default=12, help="chunks per batched call")
    ap.add_argument("--repeat", type=int, default=3, help="passes over the map; the last one is reported (the first one beside it)")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    sizes = chunk_sizes(args.small)
    merged, dt_first, dt = run_map(sizes, world, rank, local, args.in_flight, dist, dev, args.batch, args.repeat)
    if world > 1:
        t = torch.tensor([dt, dt_first], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_first = float(t[0].item()), float(t[1].item())
    if rank == 0:
        assert sorted(merged) == list(range(len(sizes))) and all(merged[i].shape[0] == sizes[i] for i in merged)
        print(json.dumps({"config": "cfg3 synthetic map", "chunks": len(sizes), "points": int(sum(sizes)), "n_gpus": world,
                          "seconds": dt, "chunks_per_s": len(sizes) / dt, "points_per_s": sum(sizes) / dt, "seconds_first_pass": dt_first,
                          "groups_total": int(sum(int(m.max()) + 1 for m in merged.values()))}), flush=True)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
