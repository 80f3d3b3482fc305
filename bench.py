#!/usr/bin/env python3
"""Benchmark of the NCuts hot path on MI355X (contract in the project brief, section (4)).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of chunks per GPU: `--in-flight` (4) host
threads, each with its own context / HIP streams, each pushing `--batch` (6) independent chunks
through ONE batched call (the chunks are the root segments of one frontier and share every kernel
launch): affinity build (TARL + spatial) from inputs already resident in HBM, recursive normalized
cut, labels back on the host, and (N > 1) the gather of the label arrays to rank 0.  One chunk
alone is latency-bound (thousands of dependent launches on small frontiers); its latency is
reported next to the throughput, and the per-chunk counters / roofline come from that single run.  Workload = BASELINE.json configs[1]: a 200 000-point
chunk, alpha = 1, theta = 0.5 (96-d features), T = 0.03; synthetic surface chunk.
Chunks are independent, so ranks process different chunks with no data-path collective
("weak" scaling: the same number of chunks per rank per step).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the fused Lanczos SpMV,
`k_lz_spmv_x`): algorithmic bytes of its launches / their summed duration, both from a profiled
repeat of one batched call with HIP start/stop events on every dispatch (library's stream).
`cpu_baseline` is the oracle (NumPy / SciPy restatement, scipy eigsh shift-invert as the
reference) timed on this host on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_POINTS = 200_000
CFG = dict(alpha=1.0, theta=0.5, gamma=0.0, T=0.03)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def spmv_bytes(rows: int, nnz: int, launches: int) -> float:
    """Algorithmic bytes of the fused SpMV launches.

    Per stored entry: 4 B column index + 8 B scaled weight.  Per row: 4 B row pointer, and
    8 B each for R_j (read once: the gathers re-use it from cache), sinv2 and the z written.
    """
    return nnz * 12.0 + rows * (4.0 + 3 * 8.0) + launches * 4.0


def pmc_traffic(batch: int):
    """HBM-side bytes per SpMV launch from the committed PMC passes (`profiles/r01_pmc_traffic.json`:
    rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of the same batched call, corrected
    as MI355X_MICROARCH.md prescribes).  Counters cannot be read inside this process, so the number is
    only reported when it was measured for the same chunks-per-batch; otherwise null."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            r = json.load(f)
    except (OSError, ValueError):
        return None, None
    if f" {batch} 1" not in r.get("command", ""):
        return None, None
    return r.get("traffic_bytes_per_launch"), "profiles/r01_pmc_traffic_summary.txt"


def cpu_baseline(seconds_budget: float = 30.0):
    """Oracle on a bounded sample: one 20 000-point chunk of the same generator and config."""
    from autoinst_amd import synth
    from oracle import ncuts_ref
    n = 20_000
    ch = synth.synthetic_chunk(n, seed=0, tarl=True)
    t0 = time.perf_counter()
    groups = ncuts_ref.ncuts(ch["points"], ch["tarl"], alpha=CFG["alpha"], theta=CFG["theta"], gamma=CFG["gamma"],
                             T=CFG["T"], fast=True)
    dt = time.perf_counter() - t0
    # linear scaling in points flatters the CPU (eigsh grows faster than N: BASELINE.md section 2)
    return {
        "value": (n / dt) / N_POINTS,
        "unit": "chunks/sec",
        "cores": 1,
        "kind": "port",
        "sample": f"oracle/ncuts_ref.ncuts (cKDTree affinity + scipy eigsh sigma=1e-10 recursion) on one {n}-point "
                  f"TARL+Spatial chunk: {dt:.1f} s, {len(groups)} groups; scaled linearly in points to a {N_POINTS}-point chunk",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--in-flight", type=int, default=4, help="host threads (contexts) per GPU")
    ap.add_argument("--batch", type=int, default=6, help="chunks per batched call (root segments of one frontier)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from autoinst_amd import ncuts_api as api
    from autoinst_amd import sharding, synth

    K = max(1, args.in_flight)   # host threads per rank, each with its own context (HIP streams + workspace)
    B = max(1, args.batch)       # chunks per ai_ncut_batch call: root segments of one frontier
    dev = torch.device("cuda", local_rank)
    ctxs = [api.Context(local_rank) for _ in range(K)]
    data = []
    # K*B different chunks (seeds 0..K*B-1), all resident in HBM.  Every rank works on the SAME set, so
    # the per-GPU work is exactly fixed as N grows (weak scaling); one chunk costs 26-59 ms depending
    # on its seed, and a real map's spread is what sharding.lpt_assign balances.
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=4) as gen:   # set-up only: ~2 s of NumPy per chunk
        for ch in gen.map(lambda i: synth.synthetic_chunk(N_POINTS, seed=i, tarl=True), range(K * B)):
            data.append((torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev)))
    torch.cuda.synchronize()

    def one_batch(k, profile=False, only_first=False):
        mine = data[k * B:(k + 1) * B][: 1 if only_first else B]
        graphs = [api.build_affinity(p, f, alpha=CFG["alpha"], theta=CFG["theta"], gamma=CFG["gamma"], ctx=ctxs[k]) for p, f in mine]
        try:
            if len(graphs) == 1:
                lab, ng, st = api.ncuts_labels(graphs[0], N_POINTS, CFG["T"], time_spmv=profile)
                labs, ngs = [lab], [ng]
            else:
                labs, ngs, st = api.ncuts_labels_batch(graphs, None, CFG["T"], time_spmv=profile)
        finally:
            nnz = graphs[0].nnz
            for g in graphs:
                g.free()
        return labs, ngs, st, nnz

    import queue
    import threading
    jobs = [queue.Queue() for _ in range(K)]   # per worker: number of batches to run (None = exit)
    qs = [queue.Queue() for _ in range(K)]     # per worker: results, one per batch

    def worker(k):
        # the K host threads live for the whole run (warm-up, timed steps, latency and profile passes)
        while True:
            n = jobs[k].get()
            if n is None:
                return
            for _ in range(n):
                try:
                    qs[k].put(one_batch(k))
                except BaseException as e:  # surface the failure in the consuming thread
                    qs[k].put(e)
                    break

    workers = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(K)]
    for t in workers:
        t.start()

    def run_steps(nsteps):
        """`nsteps` steps: every host thread pushes `nsteps` batches back to back (the library calls release
        the GIL, so the K batches really are in flight together, and a thread does not wait for the others
        between steps); this thread takes each step's K results as they complete and (N > 1) gathers that
        step's label arrays to rank 0.  Returns the last step's results."""
        for k in range(K):
            jobs[k].put(nsteps)
        last = None
        for _ in range(nsteps):
            res = [q.get() for q in qs]
            for r in res:
                if isinstance(r, BaseException):
                    raise r
            local = {(rank * K + k) * B + b: res[k][0][b] for k in range(K) for b in range(B)}
            merged = sharding.gather_labels(local, device=dev) if world > 1 else local
            last = (res, merged)
        return last

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    last = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    res, merged = last
    nnz = res[0][3]
    stb = res[0][2]

    # one chunk alone (latency and per-chunk counters), then a profiled repeat of one batched call:
    # HIP start/stop events on every SpMV dispatch
    one_batch(0, only_first=True)
    t1 = time.perf_counter()
    for _ in range(3):
        labs1, ngs1, st, _ = one_batch(0, only_first=True)
    latency_ms = 1e3 * (time.perf_counter() - t1) / 3
    ng = ngs1[0]
    _, _, stp, _ = one_batch(0, profile=True)   # the batched call as timed above, SpMV dispatches bracketed
    barrier()
    for k in range(K):
        jobs[k].put(None)
    for t in workers:
        t.join()

    free_b, total_b = torch.cuda.mem_get_info(dev)
    if rank == 0:
        assert merged is not None and len(merged) == world * K * B and all(v.shape[0] == N_POINTS for v in merged.values())
        launches = int(stp["lanczos_steps"])
        b = spmv_bytes(int(stp["spmv_rows"]), int(stp["spmv_nnz"]), launches)
        ach = b / (stp["ms_spmv"] * 1e-3) / 1e9 if stp["ms_spmv"] > 0 else 0.0
        traffic, traffic_src = pmc_traffic(B)
        out = {
            "metric": "chunks/sec (200k-pt TARL+Spatial NCuts chunk: affinity build + recursive normalized cut)",
            "value": world * K * B * args.steps / elapsed,
            "unit": "chunks/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "configs[1]: single 200k-point surface chunk, TARL(96-d)+Spatial affinities, "
                                   f"alpha=1 theta=0.5 T=0.03; per GPU per step {K} host threads x {B} chunks batched into one frontier",
                       "n_points": N_POINTS, "nnz": int(nnz), "chunks_per_step": world * K * B, "threads_per_gpu": K, "chunks_per_batch": B,
                       "parallelism": f"chunk-dp{world}"},
            "hbm_in_use_gb": (total_b - free_b) / 1e9,   # inputs + K workspaces (arena, cached graph buffers) + torch
            "single_chunk_latency_ms": latency_ms,
            "batch_ncut_ms": stb["ms_total"],
            "eigensolve_ms": st["ms_eigen"],
            "ncut_ms": st["ms_total"],
            "sweep_ms": st["ms_sweep"],
            "rebuild_ms": st["ms_rebuild"],
            "levels": int(st["levels"]),
            "lanczos_solves": int(st["lanczos_solves"]),
            "null_solves": int(st["null_solves"]),
            "lanczos_steps": int(st["lanczos_steps"]),
            "groups": int(ng),
            "unconverged": int(st["unconverged"]),
            "roofline": {
                "kernel": "k_lz_spmv_x",
                "bound": "hbm",
                "achieved": ach,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "launches": launches,
                "avg_launch_us": 1e3 * stp["ms_spmv"] / max(launches, 1),
                "bytes_per_launch_avg": b / max(launches, 1),
            },
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
