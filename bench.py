#!/usr/bin/env python3
"""Benchmark of the NCuts hot path on MI355X (contract in the project brief, section (4)).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks
itself (one child process per GPU, started BEFORE this process imports torch or touches HIP; the
parent only waits, relays rank 0's JSON line and returns non-zero if any rank failed).

One "step" = one pass of the hot path over one batch of chunks per GPU: `--batches` (2) batched calls of
`--batch` (12) independent chunks each (the connected segments of a call's chunks iterate in one pool, each at
its own Lanczos step count, and share every kernel launch), taken from one queue by `--in-flight` (2) host
threads with their own context / HIP streams: affinity build (TARL + spatial) from inputs already resident in HBM, recursive normalized
cut, labels back on the host, and (N > 1) the gather of the label arrays to rank 0.  Workload =
BASELINE.json configs[1]: a 200 000-point chunk, alpha = 1, theta = 0.5 (96-d features), T = 0.03;
synthetic surface chunk.  Chunks are independent (reference `pipeline/run_pipeline.py:160-179`),
so the world's chunk list is dealt to the ranks by `sharding.lpt_assign` and there is no data-path
collective ("weak" scaling: the same number of chunks per rank per step).

Rank 0 prints ONE JSON line.
* `roofline` is for the dominant kernel (the fused Lanczos SpMV, `fk_spmv`): algorithmic bytes of its
  launches / their summed duration, measured live: every launch stamps its own span on the device clock (this
  agrees with rocprofv3's per-kernel average).  Two regimes are measured and both are reported: `frac_solo` (one batched call alone on
  the device) and `frac_overlapped` (the `--in-flight` host threads in flight together, i.e. the regime
  `value` is quoted in); `frac` = the overlapped one.  `frac_aggregate` = all SpMV bytes of a step / the
  step's wall time.  The rocprofv3 summaries of both regimes are tracked under `profiles/`.
* `cpu_baseline` is the oracle (NumPy / SciPy restatement, scipy eigsh shift-invert as the reference):
  `value` = the MEASURED single-process run on the full 200k chunk (cached in `profiles/`, it takes hours; `cached_host`
  says where it ran), `pool` = a process pool of min(host cores, 128) workers on THIS host over 20k-26k-point chunks,
  timed now, and one single-process 50k-point run on this host (~2-3 min; skipped when the pool leg took more than 200 s
  unless `--cpu-50k`, never with `--no-cpu-50k`).
* `value_host_inputs` = the same loop with the inputs in pinned host memory, sent on a copy stream per host thread
  one batch ahead of its kernels; `value` has them resident in HBM.
* `single_chunk_latency_ms` and the per-chunk counters are those of the seed-0 chunk alone; `single_chunk_latency_ms_all`
  / `lanczos_steps_all` = mean / min / max over all chunks of the step, each alone.  `roofline.device_copy_gbps` = a
  plain device-to-device copy on the same box.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_POINTS = 200_000
CFG = dict(alpha=1.0, theta=0.5, gamma=0.0, T=0.03)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CPU_200K_CACHE = os.path.join(ROOT, "profiles", "r02_cpu_oracle_200k.json")
PMC_TRAFFIC = [os.path.join(ROOT, "profiles", f"r0{r}_pmc_traffic.json") for r in (5, 4, 3, 2)]   # newest first


def spmv_bytes(rows: int, nnz: int, launches: int) -> float:
    """Algorithmic bytes of the fused SpMV launches.

    Per stored entry: 4 B column index + 8 B scaled weight.  Per row: 4 B row pointer, and
    8 B each for R_j (read once: the gathers re-use it from cache), sinv2 and the z written.
    """
    return nnz * 12.0 + rows * (4.0 + 3 * 8.0) + launches * 4.0


def pmc_traffic(batch: int):
    """HBM-side bytes per SpMV launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE and --pmc
    WRITE_SIZE in separate runs of the same batched call, corrected as MI355X_MICROARCH.md prescribes).
    Counters cannot be read inside this process, so the number is only reported when it was measured for
    the same chunks-per-batch; otherwise null."""
    for path in PMC_TRAFFIC:
        try:
            with open(path) as f:
                r = json.load(f)
        except (OSError, ValueError):
            continue
        if f" {batch} 1" not in r.get("command", ""):
            continue
        return r.get("traffic_bytes_per_launch"), os.path.relpath(path, ROOT)
    return None, None


# ----------------------------------------------------------------------------- CPU baseline
def _cpu_pool_worker(job):
    """One oracle run in a pool worker (1 BLAS thread): returns (n, seconds, groups)."""
    n, seed = job
    os.environ["OMP_NUM_THREADS"] = "1"
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    from autoinst_amd import synth
    from oracle import ncuts_ref
    ch = synth.synthetic_chunk(n, seed=seed, tarl=True)
    t0 = time.perf_counter()
    groups = ncuts_ref.ncuts(ch["points"], ch["tarl"], alpha=CFG["alpha"], theta=CFG["theta"], gamma=CFG["gamma"],
                             T=CFG["T"], fast=True)
    return n, time.perf_counter() - t0, len(groups)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_one_worker(job):
    return _cpu_pool_worker(job)


def _cpu_device_algorithm_worker(job):
    """The DEVICE's algorithm (Lanczos on I - L_sym with u1 projected out, component split, the reference's sweep and recursion:
    tests/gpu_model.py, the NumPy model the GPU suite holds the HIP path to) on one full-size chunk, one core: returns
    (n, affinity seconds, cut seconds, groups, canonical labels)."""
    n, seed = job
    os.environ["OMP_NUM_THREADS"] = "1"
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpu_model
    from autoinst_amd import synth
    from oracle import ncuts_ref
    ch = synth.synthetic_chunk(n, seed=seed, tarl=True)
    t0 = time.perf_counter()
    A = ncuts_ref.affinity_sparse(ch["points"], ch["tarl"], alpha=CFG["alpha"], theta=CFG["theta"], gamma=CFG["gamma"])
    t1 = time.perf_counter()
    sys.setrecursionlimit(10000)
    groups = gpu_model.normalized_cut_model(A, n, np.arange(n), T=CFG["T"])
    t2 = time.perf_counter()
    return n, t1 - t0, t2 - t1, len(groups), ncuts_ref.groups_to_labels(groups, n).astype(np.int32)


def cpu_baseline(workers: int | None = None, with_50k: bool | None = None, gpu_labels=None, gpu_seed: int = 0):
    """The oracle beside the GPU number.

    (i) `value`: the single-process oracle on the real 200k chunk, measured once in the build container by
    `oracle/gen_fullsize.py` and cached (`profiles/r02_cpu_oracle_200k.json`; it takes hours, which a bench run
    cannot afford).  (ii) `pool`: a process pool of this host's cores (1 BLAS thread each, one chunk of
    20k-26k points -- the real chunk-size range -- per worker), timed now, ~30 s.
    """
    import multiprocessing as mp
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if workers is None:
        # the brief's target is quoted against a 128-core CPU run of the reference (run_pipeline.py:160-179 as a process pool)
        workers = int(os.environ.get("AI_BENCH_CPU_WORKERS", min(avail, 128)))
    workers = max(1, workers)
    sizes = [int(x) for x in np.linspace(20_000, 26_000, workers)]
    jobs = [(n, i) for i, n in enumerate(sizes)]
    ctx = mp.get_context("spawn")   # fresh interpreters: nothing of this process's HIP state is inherited
    t0 = time.perf_counter()
    with ctx.Pool(workers) as pool:
        res = pool.map(_cpu_pool_worker, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    same_box = None
    if with_50k is None:   # default: yes, unless the pool leg already took long (a slow or heavily shared host)
        with_50k = wall <= 200.0
    if with_50k:   # one single-process run at 50k points on THIS host (the oracle grows faster than linearly in N)
        with ctx.Pool(1) as pool:
            n50, sec50, g50 = pool.map(_cpu_one_worker, [(50_000, 0)])[0]
        same_box = {"n": n50, "seconds": sec50, "groups": g50, "chunks_per_sec": 1.0 / sec50, "cpu_model": _cpu_model()}
    # The full 200k-point chunk on THIS host, in seconds instead of hours: the device's own algorithm on one core (the reference's
    # shift-invert path needs 2.1 h for the same chunk: `value` below).  Its partition is compared with the labels the GPU returned
    # for the same chunk in this run: a full-size parity check inside the bench.
    dev_algo = None
    try:
        with ctx.Pool(1) as pool:
            n2, sa, sc, g2, lab2 = pool.map(_cpu_device_algorithm_worker, [(N_POINTS, gpu_seed)])[0]
        dev_algo = {"n": n2, "affinity_seconds": sa, "cut_seconds": sc, "seconds": sa + sc, "chunks_per_sec": 1.0 / (sa + sc), "groups": g2, "cores": 1,
                    "cpu_model": _cpu_model(), "what": "tests/gpu_model.py: the device's algorithm in NumPy / SciPy (Lanczos without re-orthogonalisation, "
                    "component split, the reference's sweep and recursion) on the seed-%d chunk; NOT the reference's shift-invert eigsh path" % gpu_seed}
        if gpu_labels is not None:
            from oracle import ncuts_ref
            dev_algo["partition_equals_gpu_labels"] = bool(ncuts_ref.partitions_equal(lab2, np.asarray(gpu_labels)))
    except Exception as e:  # noqa: BLE001 -- a reported leg, not the measurement
        dev_algo = {"error": f"{type(e).__name__}: {e}"}
    pts = sum(r[0] for r in res)
    pool_info = {
        "cores": workers, "host_cores_available": avail, "cpu_model": _cpu_model(),
        "chunks": len(res), "points": pts, "wall_seconds": wall,
        "chunks_per_sec_small": len(res) / wall,            # chunks of 20k-26k points per second, all workers
        "points_per_sec": pts / wall,
        "per_chunk_seconds_min_max": [min(r[1] for r in res), max(r[1] for r in res)],
        "sample": f"{workers} workers x 1 chunk of {sizes[0]}-{sizes[-1]} points (seeds 0..{workers - 1}), oracle/ncuts_ref.ncuts, 1 BLAS thread each",
    }
    out = {"unit": "chunks/sec", "kind": "port", "pool": pool_info, "single_process_50k_this_host": same_box,
           "device_algorithm_200k_this_host": dev_algo}
    try:
        with open(CPU_200K_CACHE) as f:
            c = json.load(f)
        sec = float(c["affinity_seconds"]) + float(c["normalized_cut_seconds"])
        out.update({
            "value": 1.0 / sec, "cores": 1, "measured": True,
            "cached_host": {"cpu_model": c["cpu_model"], "nproc": c["nproc"], "concurrent_jobs": c.get("concurrent_jobs", 1),
                            "note": "the build container, not this host"},
            "sample": f"oracle/gen_fullsize.py on the full {c['n']}-point TARL+Spatial chunk (seed {c['seed']}), single process, "
                      f"measured {sec:.0f} s ({c['eigsh_calls']} eigsh calls, {c['groups']} groups) on {c['cpu_model']} "
                      f"({c['nproc']} cores, {c.get('concurrent_jobs', 1)} such jobs running side by side); cached in "
                      f"{os.path.relpath(CPU_200K_CACHE, ROOT)}",
        })
    except (OSError, ValueError, KeyError):
        # no cached full-size run: fall back to the pool's point rate, flagged as an extrapolation
        out.update({"value": pool_info["points_per_sec"] / N_POINTS, "cores": workers, "measured": False, "extrapolated": True,
                    "sample": pool_info["sample"] + f"; points/s scaled linearly to {N_POINTS}-point chunks (flatters the CPU)"})
    return out


# ----------------------------------------------------------------------------- host placement (host-input leg)
def _parse_cpulist(txt: str):
    cpus = set()
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_cpus(torch, index: int):
    """(NUMA node of GPU `index`, CPUs of that node this process may run on, note).  The pinned staging blocks of the host-input
    leg are first touched, and the loader thread that sends them runs, on those CPUs: pinned memory on the far socket halves
    the H2D rate and the copy then no longer hides behind the kernels (round 3: 0.98 of the resident rate on the builder's
    boxes, 0.795 on the driver's)."""
    allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set(range(os.cpu_count() or 1))
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip())
    except (OSError, ValueError, AttributeError) as e:
        return None, set(allowed), f"GPU NUMA node unknown ({type(e).__name__})"
    if node < 0:
        return node, set(allowed), "the platform reports no NUMA node for the GPU"
    try:
        node_cpus = _parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
    except OSError:
        return node, set(allowed), "node cpulist unreadable"
    near = node_cpus & set(allowed)
    if not near:
        return node, set(allowed), f"none of this process's {len(allowed)} CPUs is on the GPU's node {node}"
    return node, near, f"{len(near)} of this process's {len(allowed)} CPUs are on the GPU's node {node}"


# ----------------------------------------------------------------------------- rank start-up (multi-GPU runs)
def _cpulist(cpus) -> str:
    """{0,1,2,3,8,9} -> '0-3,8-9'"""
    out, run = [], []
    for c in sorted(cpus):
        if run and c == run[-1] + 1:
            run.append(c)
        else:
            if run:
                out.append(run)
            run = [c]
    if run:
        out.append(run)
    return ",".join(str(r[0]) if len(r) == 1 else f"{r[0]}-{r[-1]}" for r in out)


def rank_startup_info(rank: int, local_rank: int, torch=None, numa=None) -> dict:
    """What a rank is running on, gathered into rank 0's JSON line (`ranks`): the first thing to look at when an N-GPU run is slow or
    hangs -- which device, which NUMA node, which CPUs each rank got."""
    import socket
    cpus = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set(range(os.cpu_count() or 1))
    info = {"rank": rank, "local_rank": local_rank, "pid": os.getpid(), "host": socket.gethostname(), "cpus": len(cpus), "cpu_list": _cpulist(cpus)}
    if torch is not None and torch.cuda.is_available():
        try:
            pr = torch.cuda.get_device_properties(local_rank)
            info.update({"device": pr.name, "gcn_arch": getattr(pr, "gcnArchName", None), "hbm_gb": round(pr.total_memory / 2 ** 30, 1),
                         "pci": "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)})
        except Exception as e:  # noqa: BLE001 -- diagnosis only
            info["device_error"] = f"{type(e).__name__}: {e}"
    if numa is not None:
        info.update({"gpu_numa_node": numa[0], "numa_note": numa[2]})
    return info


def first_collective(dist, torch, device, rank: int, world: int, limit_s: float, what: str) -> float:
    """The first collective of the run, bounded: an 8-byte all_reduce that must return within `limit_s` seconds.  More than one
    rank has never met RCCL on this pool (the builder has one GPU), so the first exchange is where a wrong rendezvous, a GPU that is
    not there or an IPC problem would show -- as a hang.  A watchdog thread turns that into a message that names the rank and a
    non-zero exit (the launcher -- `self_launch` or torchrun -- then ends the other ranks); nothing is re-executed.  Returns ms."""
    import threading
    done = threading.Event()

    def watchdog():
        if not done.wait(limit_s):
            sys.stderr.write(f"bench.py: rank {rank} of {world} (pid {os.getpid()}): the first {what} collective (8-byte all_reduce) did not complete "
                             f"within {limit_s:.0f} s -- rendezvous at {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}; "
                             "a rank that is missing, a GPU that is not visible or HSA_ENABLE_IPC_MODE_LEGACY != 0 are the usual causes\n")
            sys.stderr.flush()
            os._exit(75)

    threading.Thread(target=watchdog, daemon=True, name="first-collective-watchdog").start()
    t0 = time.perf_counter()
    t = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(t)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    got = int(t.item())
    done.set()
    if got != world:
        raise SystemExit(f"bench.py: rank {rank}: the first all_reduce returned {got}, expected the world size {world}")
    return 1e3 * (time.perf_counter() - t0)


def gather_rank_infos(dist, info: dict, world: int, group=None):
    """every rank's start-up record on every rank (object collective on a gloo group: control plane, not the data path)"""
    out = [None] * world
    dist.all_gather_object(out, info, group=group)
    return out


# ----------------------------------------------------------------------------- self-launch
def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n: int, argv) -> int:
    """Start `n` ranks of this script as child processes (one per GPU) and relay rank 0's output.

    Runs before torch is imported: the parent never touches HIP, it only waits.  If a rank exits non-zero
    the others are terminated (by their PIDs) and the code is passed on.
    """
    port = int(os.environ.get("MASTER_PORT", "0")) or _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    code = 0
    try:
        while True:
            rcs = [p.poll() for p in procs]
            bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad:
                code = bad[0][1]
                sys.stderr.write("bench.py: " + ", ".join(f"rank {r} exited with code {rc}" for r, rc in bad) + f" (of {n} ranks); ending the others\n")
                break
            if all(rc == 0 for rc in rcs):
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    out0 = b"".join(c for c in chunks if c)
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    return code


# ----------------------------------------------------------------------------- dry run (CPU, gloo)
def dry_run(args, world, rank):
    """The multi-rank plumbing without a GPU: LPT deal of an uneven chunk list, per-step gather of label
    arrays to rank 0 over `gloo`, barrier + max-over-ranks timing, one JSON line.  Computes nothing."""
    import torch
    import torch.distributed as dist
    from autoinst_amd import sharding
    infos, first_ms = [rank_startup_info(rank, int(os.environ.get("LOCAL_RANK", "0")))], None
    if world > 1:
        import datetime
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
        first_ms = first_collective(dist, torch, torch.device("cpu"), rank, world, float(os.environ.get("AI_BENCH_FIRST_COLLECTIVE_S", "120")), "gloo")
        infos = gather_rank_infos(dist, infos[0], world)
    rng = np.random.default_rng(0)
    nchunks = world * args.in_flight * args.batch
    sizes = np.exp(rng.uniform(np.log(3_000), np.log(30_000), nchunks)).astype(int).tolist()   # cfg3's chunk-size mix: an UNEVEN deal
    deal = sharding.lpt_assign(sizes, world)
    mine = deal[rank]
    loads = [sum(sharding.chunk_cost(sizes[c]) for c in d) for d in deal]
    t_pack = t_gather = t_parse = 0.0

    def step():
        nonlocal t_pack, t_gather, t_parse
        t0 = time.perf_counter()
        local = {c: (np.arange(sizes[c], dtype=np.int32) % 7) for c in mine}
        t1 = time.perf_counter()
        if world == 1:
            return local
        fut = sharding.gather_labels_async(local, device=torch.device("cpu"))
        t2 = time.perf_counter()
        out = None if fut is None else fut.result()
        t3 = time.perf_counter()
        t_pack += t1 - t0
        t_gather += t2 - t1
        t_parse += t3 - t2
        return out

    for _ in range(args.warmup):
        step()
    t_pack = t_gather = t_parse = 0.0
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        merged = step()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        assert merged is not None and sorted(merged) == list(range(nchunks)) and all(merged[c].shape[0] == sizes[c] for c in merged)
        mean_load = sum(loads) / world
        print(json.dumps({"metric": "dry run: chunk deal + label gather only (no compute)", "dry": True, "value": nchunks * args.steps / elapsed,
                          "unit": "chunks/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "int32", "data": "synthetic", "rccl_ranks": world, "backend": "gloo" if world > 1 else "none",
                          "config": {"workload": "label arrays of an uneven chunk list (3k-30k points, log-uniform), LPT-dealt", "chunks_per_step": nchunks},
                          # what the deal and the root cost, in numbers: the slowest rank's share of the cost model over the mean share
                          # (1.0 = perfectly even: the step time of a real run is bound by it), and the root's per-step times
                          "lpt": {"points_per_rank": [sum(sizes[c] for c in d) for d in deal], "chunks_per_rank": [len(d) for d in deal],
                                  "cost_max_over_mean": max(loads) / mean_load, "predicted_efficiency_from_imbalance": mean_load / max(loads)},
                          "root_ms_per_step": {"pack": 1e3 * t_pack / args.steps, "exchange_calls": 1e3 * t_gather / args.steps,
                                               "wait_for_parse": 1e3 * t_parse / args.steps},
                          "label_bytes_per_step": int(4 * sum(sizes)), "ranks": infos, "first_collective_ms": first_ms}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--in-flight", type=int, default=2, help="host threads (contexts) per GPU")
    ap.add_argument("--batch", type=int, default=12, help="chunks per batched call (their segments iterate in one pool)")
    ap.add_argument("--cpu-50k", action="store_true", help="cpu_baseline: the single-process 50k-point oracle run on this host (~3 min) even when the pool leg took more than 200 s")
    ap.add_argument("--no-cpu-50k", action="store_true", help="cpu_baseline: skip the single-process 50k-point oracle run on this host")
    ap.add_argument("--batches", type=int, default=0, help="batched calls per GPU per step (default: one per host thread)")
    ap.add_argument("--builders", type=int, default=1, help="host threads (one context each) that build the affinity graphs of the next batches while the --in-flight threads run the batched cuts (0: every thread builds its own batch's graphs first, as until round 4)")
    ap.add_argument("--window-chunks", type=int, default=0, help="admission window of a batched call in chunks (0: the library's default, 4.8 M rows = 24 chunks): chunks beyond it wait inside the call and join as earlier ones finish")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-inputs", action="store_true", help="skip the pinned-host-input leg (value_host_inputs)")
    ap.add_argument("--no-latency-all", action="store_true", help="skip the one-chunk-alone pass over all chunks of the step (profiled runs)")
    ap.add_argument("--dry", action="store_true", help="CPU-only rehearsal of the multi-rank plumbing (gloo), no compute")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # nothing GPU-related has been imported yet: the ranks are fresh child processes
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE = {world} but --gpus {args.gpus}: launch one rank per GPU")
    if args.dry:
        return dry_run(args, world, rank)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        # one rank per GPU on one node: keep the rank's threads (two feeding threads, their helper threads, the gather's parser) and
        # the pinned memory they first-touch on the CPUs of the GPU's NUMA node
        _numa = None
        try:
            _numa = gpu_numa_cpus(torch, local_rank)
            if _numa[1] and len(_numa[1]) >= 8:
                os.sched_setaffinity(0, _numa[1])
        except OSError:
            pass
        import datetime
        # a rank that dies leaves the others in a collective: the timeout turns that into a failure
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=600))
        # control plane on gloo (start-up records); then the first RCCL collective, bounded and named (first_collective)
        ctl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=300))
        rank_infos = gather_rank_infos(dist, rank_startup_info(rank, local_rank, torch, _numa), world, group=ctl)
        first_ms = first_collective(dist, torch, torch.device("cuda", local_rank), rank, world, float(os.environ.get("AI_BENCH_FIRST_COLLECTIVE_S", "180")), "RCCL")
        if rank == 0:
            for ri in rank_infos:
                sys.stderr.write("bench.py: " + json.dumps(ri) + "\n")
            sys.stderr.write(f"bench.py: first RCCL all_reduce over {world} ranks: {first_ms:.1f} ms\n")
    else:
        rank_infos, first_ms = [rank_startup_info(rank, local_rank, torch)], None

    from autoinst_amd import ncuts_api as api
    from autoinst_amd import sharding, synth

    K = max(1, args.in_flight)   # host threads per rank, each with its own context (HIP streams + workspace)
    B = max(1, args.batch)       # chunks per ai_ncut_batch call: root segments of one frontier
    M = args.batches if args.batches > 0 else K   # batched calls per step, taken from one queue by the K threads
    dev = torch.device("cuda", local_rank)
    ctxs = [api.Context(local_rank) for _ in range(K)]
    NB = max(0, args.builders)   # builder threads: the graphs of batch k + 1 are built (own context, own stream) while batch k is cut
    bctxs = [api.Context(local_rank) for _ in range(NB)]
    # The world's chunk list (world*M*B chunks of N_POINTS points) is dealt to the ranks by the LPT rule the
    # map driver uses; chunk c's input is the synthetic chunk of seed c mod M*B, so every rank holds the same
    # M*B different inputs resident in HBM and the per-GPU work is exactly fixed as N grows (weak scaling;
    # one chunk costs 26-59 ms depending on its seed).
    my_chunks = sharding.lpt_assign([N_POINTS] * (world * M * B), world)[rank]
    assert len(my_chunks) == M * B
    data, host = [], []
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=4) as gen:   # set-up only: ~2 s of NumPy per chunk
        for ch in gen.map(lambda c: synth.synthetic_chunk(N_POINTS, seed=c % (M * B), tarl=True), my_chunks):
            data.append((torch.from_numpy(ch["points"]).to(dev), torch.from_numpy(ch["tarl"]).to(dev)))
            if not args.no_host_inputs and world == 1:   # the host-input leg is an N = 1 measurement
                host.append((ch["points"], ch["tarl"]))
    torch.cuda.synchronize()

    # host-input leg: each batch's inputs lie in ONE pinned block (points of the B chunks, then their features).  One loader
    # thread sends the batches in queue order, one copy each on its own stream, into a ring of K+1 device staging blocks and
    # hands each staged batch to whichever host thread is free next, so the transfer of the following batch runs beside the
    # kernels of the K batches in flight.  (tools/probe_h2d_interference.py: copies queued back to back on a stream cost the
    # kernels beside them 27 %, one copy at a time 1-4 %.)
    host_blocks, stage_blocks, views = [], [], None
    numa_node, near_cpus, numa_note = None, None, None
    h2d = {"bytes": 0, "seconds": 0.0, "copies": 0, "wait_block_seconds": 0.0}
    if host:
        np_, nf_ = host[0][0].size, host[0][1].size
        numa_node, near_cpus, numa_note = gpu_numa_cpus(torch, local_rank)
        all_cpus = os.sched_getaffinity(0)
        os.sched_setaffinity(0, near_cpus)   # first touch of the pinned blocks on the GPU's node (this thread only; restored below)
        for k in range(M):
            blk = torch.empty(B * (np_ + nf_), dtype=torch.float64).pin_memory()
            for b, (p, f) in enumerate(host[k * B:(k + 1) * B]):
                blk[b * np_:(b + 1) * np_] = torch.from_numpy(p).reshape(-1)
                blk[B * np_ + b * nf_:B * np_ + (b + 1) * nf_] = torch.from_numpy(f).reshape(-1)
            host_blocks.append(blk)
        os.sched_setaffinity(0, all_cpus)
        shp_p, shp_f = host[0][0].shape, host[0][1].shape
        host = True
        stage_blocks = [torch.empty(B * (np_ + nf_), dtype=torch.float64, device=dev) for _ in range(K + 1)]
        views = lambda d: [(d[b * np_:(b + 1) * np_].view(shp_p), d[B * np_ + b * nf_:B * np_ + (b + 1) * nf_].view(shp_f)) for b in range(B)]

    # (int += under the GIL)
    guard_counts = {"restarted_solves": 0, "hist_retries": 0, "accepted_above_limit": 0, "check_timeouts": 0, "calls": 0,
                    "max_true_resid": 0.0, "max_resid_estimate": 0.0, "true_resid_limit": 0.0}

    def build_batch(k, ctx, only_first=False, from_host=False, staged=None):
        # the affinity graphs of batch k of the rank's M batches, on context `ctx`
        sl = slice(k * B, (k + 1) * B)
        if from_host:
            mine = staged      # the loader thread has put this batch into a device staging block
        else:
            mine = data[sl]
        mine = mine[: 1 if only_first else B]
        return [api.build_affinity(p, f, alpha=CFG["alpha"], theta=CFG["theta"], gamma=CFG["gamma"], ctx=ctx) for p, f in mine]

    def one_batch(k, profile=False, only_first=False, from_host=False, w=None, staged=None, graphs=None):
        # batch k of the rank's M batches, cut by host thread w on that thread's context (its graphs built here unless a builder thread has)
        ctx = ctxs[k % K if w is None else w]
        if graphs is None:
            graphs = build_batch(k, ctx, only_first, from_host, staged)
        try:
            if len(graphs) == 1:
                lab, ng, st = api.ncuts_labels(graphs[0], N_POINTS, CFG["T"], time_spmv=profile, ctx=ctx)
                labs, ngs = [lab], [ng]
            else:
                labs, ngs, st = api.ncuts_labels_batch(graphs, None, CFG["T"], time_spmv=profile, window_rows=(args.window_chunks * N_POINTS) or None, ctx=ctx)
        finally:
            nnz = graphs[0].nnz
            for g in graphs:
                g.free()
        # the library's two self-checks (0 in a healthy run; labels do not depend on them): Ritz pairs that failed the true-residual test
        # and were solved again, waves whose packed histories failed their header check and were packed again -- over every call of the run
        guard_counts["restarted_solves"] += int(st.get("restarted_solves", 0))
        guard_counts["hist_retries"] += int(st.get("hist_retries", 0))
        guard_counts["accepted_above_limit"] += int(st.get("accepted_above_limit", 0))
        guard_counts["check_timeouts"] += int(st.get("check_timeouts", 0))
        # how far from an eigenpair the cut vectors were: the largest TRUE residual ||M v - theta v|| / ||v|| of any pair that was cut
        # (measured on the segment's own entries), beside the Lanczos estimate that stopped the solve and the bar the library enforces
        guard_counts["max_true_resid"] = max(guard_counts["max_true_resid"], float(st.get("max_true_resid", 0.0)))
        guard_counts["max_resid_estimate"] = max(guard_counts["max_resid_estimate"], float(st.get("max_resid", 0.0)))
        guard_counts["true_resid_limit"] = float(st.get("true_resid_limit", 0.0))
        guard_counts["calls"] += 1
        return labs, ngs, st, nnz

    import queue
    import threading
    # The K host threads live for the whole run (warm-up, timed steps, latency and profile passes) and take batches from ONE
    # queue, as workers of a map driver take chunks: a batch costs 160-260 ms depending on its six chunks, so a fixed
    # batch-per-thread deal would leave the faster threads idle at the end of the timed region.
    work = queue.Queue()       # (step, batch, kwargs); None = exit
    results = {}               # (step, batch) -> result
    cv = threading.Condition()

    free_blocks = queue.Queue()
    for i in range(len(stage_blocks)):
        free_blocks.put(i)
    uploads = queue.Queue()    # host-input jobs on their way to `work`

    def loader():
        if near_cpus:
            os.sched_setaffinity(0, near_cpus)   # (pid 0 = the calling thread)
        cs = torch.cuda.Stream(device=dev)
        while True:
            job = uploads.get()
            if job is None:
                return
            t_w = time.perf_counter()
            i = free_blocks.get()
            t_c = time.perf_counter()
            with torch.cuda.stream(cs):
                stage_blocks[i].copy_(host_blocks[job[1]], non_blocking=True)
            cs.synchronize()
            t_e = time.perf_counter()
            h2d["bytes"] += host_blocks[job[1]].numel() * 8
            h2d["seconds"] += t_e - t_c
            h2d["wait_block_seconds"] += t_c - t_w
            h2d["copies"] += 1
            work.put(job + (i,))

    # With builder threads the jobs pass through them: `work` -> builder (graphs of the batch, on its own context) -> `built` -> worker
    # (the batched cut).  `built` holds at most K batches ahead of the cuts (12 graphs of 88 MB each per batch).
    built = queue.Queue(maxsize=max(1, K)) if NB else None

    def builder(bi):
        while True:
            job = work.get()
            if job is None:
                return
            step, k, kw = job[:3]
            try:
                if len(job) == 4:
                    try:
                        g = build_batch(k, bctxs[bi], kw.get("only_first", False), True, views(stage_blocks[job[3]]))
                    finally:
                        free_blocks.put(job[3])   # (the build has synchronised its stream: the staging block is free again)
                else:
                    g = build_batch(k, bctxs[bi], kw.get("only_first", False))
            except BaseException as e:  # surface the failure in the consuming thread
                g = e
            built.put((step, k, kw, g))

    def worker(w):
        while True:
            job = (built if NB else work).get()
            if job is None:
                return
            step, k, kw = job[:3]
            try:
                if NB:
                    if isinstance(job[3], BaseException):
                        raise job[3]
                    r = one_batch(k, w=w, graphs=job[3], **{a: b for a, b in kw.items() if a != "from_host"})
                elif len(job) == 4:
                    try:
                        r = one_batch(k, w=w, staged=views(stage_blocks[job[3]]), **kw)
                    finally:
                        free_blocks.put(job[3])
                else:
                    r = one_batch(k, w=w, **kw)
            except BaseException as e:  # surface the failure in the consuming thread
                r = e
            with cv:
                results[(step, k)] = r
                cv.notify_all()

    workers = [threading.Thread(target=worker, args=(w,), daemon=True) for w in range(K)]
    workers += [threading.Thread(target=builder, args=(bi,), daemon=True) for bi in range(NB)]
    if host:
        workers.append(threading.Thread(target=loader, daemon=True))
    for t in workers:
        t.start()

    def run_steps(nsteps, **kw):
        """`nsteps` steps = nsteps * M batches, queued at once: the library calls release the GIL, so K batches really are
        in flight together, and a thread does not wait for the others between steps.  This thread takes each step's M
        results as they complete and (N > 1) gathers that step's label arrays to rank 0.  Returns the last step's results."""
        for s_ in range(nsteps):
            for k in range(M):
                (uploads if kw.get("from_host") else work).put((s_, k, kw))
        last, pending = None, None
        for s_ in range(nsteps):
            with cv:
                cv.wait_for(lambda: all((s_, k) in results for k in range(M)))
                res = [results.pop((s_, k)) for k in range(M)]
            for r in res:
                if isinstance(r, BaseException):
                    if world > 1:
                        # the other ranks are (or will be) inside a collective: leave so that the launcher ends them
                        sys.stderr.write(f"rank {rank}: {type(r).__name__}: {r}\n")
                        sys.stderr.flush()
                        os._exit(1)
                    raise r
            local = {my_chunks[k * B + b]: res[k][0][b] for k in range(M) for b in range(B)}
            if world > 1:
                # rooted exchange issued by this thread; the root parses step s on a helper thread while step s + 1 computes
                fut = sharding.gather_labels_async(local, device=dev)
                if pending is not None:
                    pending.result()
                pending = fut
                last = (res, None)
            else:
                last = (res, local)
        if world > 1:
            last = (last[0], pending.result() if pending is not None else None)   # inside the timed region: the last step's labels are on the root
        return last

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(nsteps, **kw):
        barrier()
        t0 = time.perf_counter()
        last = run_steps(nsteps, **kw)
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, last

    # Untimed set-up of the K workspaces: a context's call workspace is collected block by block by the first call of a shape
    # and replaced by one block of the largest need at the start of the next call (`ai_arena::consolidate`, about 75 ms once).
    # Which thread takes which batch of the timed region is decided by the queue, so every context sees every batch here, and
    # one more call lets the consolidation happen outside the timed region as well.
    for w in range(K):
        for k in list(range(M)) + [0]:
            one_batch(k, w=w)
    if args.warmup > 0:
        run_steps(args.warmup)
    elapsed, last = timed(args.steps)
    res, merged = last
    nnz = res[0][3]
    stb = res[0][2]

    # ---- the same loop with the inputs in pinned host memory (the library copies them on the context's stream)
    host_steps, elapsed_host, hbm_host_leg, host_leg = 0, None, None, None
    if host:
        host_steps = args.steps   # as many as the resident leg: the first batch of each thread has nothing to hide its transfer behind
        # the link alone: one batch's block, nothing else on the device
        torch.cuda.synchronize()
        t_a = time.perf_counter()
        for _ in range(3):
            stage_blocks[0].copy_(host_blocks[0], non_blocking=True)
            torch.cuda.synchronize()
        h2d_alone_gbps = 3 * host_blocks[0].numel() * 8 / (time.perf_counter() - t_a) / 1e9
        run_steps(1, from_host=True)
        h2d.update({"bytes": 0, "seconds": 0.0, "copies": 0, "wait_block_seconds": 0.0})
        elapsed_host, _ = timed(host_steps, from_host=True)
        host_leg = {
            "gpu_numa_node": numa_node, "placement": numa_note, "loader_cpus": len(near_cpus) if near_cpus else None,
            "h2d_gbps_alone": h2d_alone_gbps,                                   # one 1.9 GB copy at a time, idle device
            "h2d_gbps_in_leg": h2d["bytes"] / max(h2d["seconds"], 1e-9) / 1e9,  # the same copies beside the kernels of the batches in flight
            "loader_copy_frac": h2d["seconds"] / elapsed_host,                  # share of the leg's wall time the loader spent copying
            "loader_wait_for_staging_block_frac": h2d["wait_block_seconds"] / elapsed_host,   # ... waiting for a free staging block (the kernels are the limit)
            "copies": h2d["copies"], "bytes_per_copy": h2d["bytes"] // max(h2d["copies"], 1),
        }
        # the leg's device staging ring (K + 1 blocks of one batch's inputs) exists only for this leg
        free_with_ring, _tot = torch.cuda.mem_get_info(dev)
        hbm_host_leg = (_tot - free_with_ring) / 1e9
        stage_blocks.clear()
        torch.cuda.empty_cache()

    # ---- roofline of the SpMV kernel, overlapped regime: the K host threads each run one batched call with HIP
    # start/stop events on every SpMV dispatch, in flight together exactly as in the timed region
    t_ov0 = time.perf_counter()
    res_ov, _ = run_steps(1, profile="clock")
    t_ov = time.perf_counter() - t_ov0
    ov = {"launches": 0, "bytes": 0.0, "ms": 0.0, "blocks": 0, "blocks_idle": 0}
    for k in range(M):
        s = res_ov[k][2]
        ov["blocks"] += int(s.get("spmv_blocks", 0))
        ov["blocks_idle"] += int(s.get("spmv_blocks_idle", 0))
        ov["launches"] += int(s["lanczos_steps"])
        ov["bytes"] += spmv_bytes(int(s["spmv_rows"]), int(s["spmv_nnz"]), int(s["lanczos_steps"]))
        ov["ms"] += s["ms_spmv"]

    # ---- one chunk alone (latency and per-chunk counters), then the solo regime: ONE batched call alone on the device
    one_batch(0, only_first=True)
    t1 = time.perf_counter()
    for _ in range(3):
        labs1, ngs1, st, _ = one_batch(0, only_first=True)
    latency_ms = 1e3 * (time.perf_counter() - t1) / 3
    ng = ngs1[0]
    # the same for every chunk of the step (the seed-0 chunk above is one of the most expensive of the 24): affinity build +
    # normalized cut of one chunk alone, once each after a warm-up call
    lat_all, steps_all = [], []
    for i in range(0 if args.no_latency_all else M * B):
        for rep in range(2):
            p_, f_ = data[i]
            t1 = time.perf_counter()
            g_ = api.build_affinity(p_, f_, alpha=CFG["alpha"], theta=CFG["theta"], gamma=CFG["gamma"], ctx=ctxs[0])
            _, _, st_i = api.ncuts_labels(g_, N_POINTS, CFG["T"])
            g_.free()
            dt_i = 1e3 * (time.perf_counter() - t1)
        lat_all.append(dt_i)
        steps_all.append(int(st_i["lanczos_steps"]))
    _, _, stp, _ = one_batch(0, profile="clock")
    # a plain device-to-device copy on the same box (quote the roofline against a measured stream number too):
    # 1 GiB read + 1 GiB written per repeat, alone on the device
    src = torch.empty(1 << 27, dtype=torch.float64, device=dev).fill_(1.0)
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    copy_gbps = 10 * 2 * src.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    torch.cuda.empty_cache()
    copy16_gbps = api.bench_copy(ctxs[0], 1 << 30, 10)   # the library's own float4 copy kernel (the guide measures 6.29 TB/s for one)
    barrier()
    for _ in range(NB if NB else K):
        work.put(None)
    if NB:
        for t in workers[K:K + NB]:
            t.join()
        for _ in range(K):
            built.put(None)
    uploads.put(None)
    for t in workers:
        t.join()

    free_b, total_b = torch.cuda.mem_get_info(dev)
    if rank == 0:
        assert merged is not None and sorted(merged) == list(range(world * M * B)) and all(v.shape[0] == N_POINTS for v in merged.values())
        launches = int(stp["lanczos_steps"])
        b = spmv_bytes(int(stp["spmv_rows"]), int(stp["spmv_nnz"]), launches)
        ach_solo = b / (stp["ms_spmv"] * 1e-3) / 1e9 if stp["ms_spmv"] > 0 else 0.0
        ach_ov = ov["bytes"] / (ov["ms"] * 1e-3) / 1e9 if ov["ms"] > 0 else 0.0
        ms_step = 1e3 * elapsed / args.steps
        ach_agg = ov["bytes"] / (ms_step * 1e-3) / 1e9   # every SpMV byte of one step / the step's wall time
        traffic, traffic_src = pmc_traffic(B)
        out = {
            "metric": "chunks/sec (200k-pt TARL+Spatial NCuts chunk: affinity build + recursive normalized cut)",
            "value": world * M * B * args.steps / elapsed,
            "unit": "chunks/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "backend": dist.get_backend() if world > 1 else "none",
            "ranks": rank_infos,                      # per rank: device, PCI id, GPU NUMA node, CPUs the rank may run on
            "first_collective_ms": first_ms,          # the bounded first RCCL all_reduce (None at N = 1)
            "config": {"workload": "configs[1]: single 200k-point surface chunk, TARL(96-d)+Spatial affinities, "
                                   f"alpha=1 theta=0.5 T=0.03; per GPU per step {M} batched calls of {B} chunks (one pool of iterating segments each) taken from a queue by {K} host threads" + (f"; {NB} builder thread(s) build the next batches' affinity graphs meanwhile" if NB else ""),
                       "n_points": N_POINTS, "nnz": int(nnz), "chunks_per_step": world * M * B, "threads_per_gpu": K, "builder_threads_per_gpu": NB, "chunks_per_batch": B, "batches_per_step": M,
                       "parallelism": f"chunk-dp{world}"},
            "value_host_inputs": (world * M * B * host_steps / elapsed_host) if elapsed_host else None,
            "host_inputs_ratio": (elapsed * host_steps / (elapsed_host * args.steps)) if elapsed_host else None,   # value_host_inputs / value
            "host_inputs_note": "same loop, each batch's points + features in one pinned host block (159 MB per chunk), sent by a loader thread with one copy per batch into a ring of three device staging blocks, one batch ahead of the kernels"
                                if elapsed_host else None,
            "host_inputs": host_leg,   # where the pinned blocks and the loader thread ran, the link's rate alone and in the leg, what the loader waited for
            "hbm_in_use_gb": (total_b - free_b) / 1e9,   # inputs + K workspaces (arena, cached graph buffers) + torch
            "hbm_in_use_host_inputs_gb": hbm_host_leg,   # the same + the host-input leg's staging ring
            "single_chunk_latency_ms": latency_ms,   # the seed-0 chunk (per-chunk counters below are its)
            "single_chunk_latency_ms_all": {"mean": sum(lat_all) / len(lat_all), "min": min(lat_all), "max": max(lat_all), "chunks": len(lat_all)} if lat_all else None,
            "lanczos_steps_all": {"mean": sum(steps_all) / len(steps_all), "min": min(steps_all), "max": max(steps_all)} if steps_all else None,
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
            "self_checks": dict(guard_counts),   # over every library call of this run (warm-up, timed, latency and profile passes)
            "roofline": {
                "kernel": "fk_spmv",
                "bound": "hbm",
                "achieved": ach_ov,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": ach_ov / HBM_PEAK_GBPS,
                "frac_is": f"overlapped: {K} host threads in flight, as in the timed region",
                "frac_overlapped": ach_ov / HBM_PEAK_GBPS,
                "frac_solo": ach_solo / HBM_PEAK_GBPS,
                "frac_aggregate": ach_agg / HBM_PEAK_GBPS,
                "device_copy_gbps": max(copy_gbps, copy16_gbps),   # the box's best plain stream rate: 1 GiB device to device (read + write bytes), alone
                "device_copy_gbps_torch": copy_gbps,             # torch's copy_
                "device_copy_gbps_float4": copy16_gbps,          # hand-written 16-byte-per-lane copy (ai_bench_copy)
                "traffic": traffic,
                "traffic_source": traffic_src,
                "overlapped": {"launches": ov["launches"], "avg_launch_us": 1e3 * ov["ms"] / max(ov["launches"], 1),
                               "bytes_per_launch_avg": ov["bytes"] / max(ov["launches"], 1), "achieved_gbps": ach_ov,
                               "step_wall_ms_while_stamping": 1e3 * t_ov,
                               # blocks dispatched for segments that had frozen since the task lists were written (they end after a record
                               # and a flag load; the lists are rewritten at the next relist)
                               "blocks": ov["blocks"], "blocks_idle": ov["blocks_idle"], "blocks_idle_frac": ov["blocks_idle"] / max(ov["blocks"], 1)},
                "solo": {"launches": launches, "avg_launch_us": 1e3 * stp["ms_spmv"] / max(launches, 1),
                         "bytes_per_launch_avg": b / max(launches, 1), "achieved_gbps": ach_solo},
                "timer": "every SpMV launch stamps its own span (first block in .. last block out) on the device clock (wall_clock64)",
                "aggregate_gbps": ach_agg,
            },
        }
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only: the other ranks would sit in the final barrier
            out["cpu_baseline"] = cpu_baseline(with_50k=False if args.no_cpu_50k else (True if args.cpu_50k else None),
                                               gpu_labels=labs1[0], gpu_seed=my_chunks[0] % (M * B))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
