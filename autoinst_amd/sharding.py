"""Chunk-parallel sharding across the GPUs of one node.

The reference runs ``for sequence in tqdm(range(#chunks)): ncuts_chunk(...)`` serially
(``pipeline/run_pipeline.py:160-179``); chunks are independent (each writes its own ``.pcd``,
``:194``), so they shard with NO data-path collective: one process per GPU, a static
longest-processing-time assignment by chunk size, and one ROOTED gather of the int32 label arrays to
rank 0 per step (counts, then a direct send of each rank's arrays to the root: a few hundred KB per chunk, RCCL over xGMI
with the ``nccl`` backend, ``gloo`` in CPU tests; the root parses on a helper thread).  ``merge_chunks_unite_instances2`` stays serial on rank 0, as in the reference.

Inside one rank `run_chunks` is that loop for one GPU: two host threads (one `Context` each) take batches of
chunks from one queue and push each batch through ONE batched call (`ncuts_labels_batch`: the connected segments of the
call's chunks iterate in one pool, each at its own step count) -- the arrangement `bench.py` measures.
"""
from __future__ import annotations

import numpy as np


def chunk_cost(n_points: int) -> float:
    """Relative cost model of one chunk: edges x recursion depth grows a little faster than N."""
    return float(n_points) ** 1.3


def lpt_assign(sizes, world_size: int):
    """Greedy LPT: chunks by decreasing cost, each to the least-loaded rank.

    Returns ``world_size`` lists of chunk indices (each ascending).  Deterministic.
    """
    if world_size <= 0:
        raise ValueError("world_size must be positive")
    order = sorted(range(len(sizes)), key=lambda i: (-chunk_cost(sizes[i]), i))
    load = [0.0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += chunk_cost(sizes[i])
    return [sorted(x) for x in out]


_parse_pool = None


def _parser():
    """One helper thread per process parses what the root has received (a step's 134 MB at 8 ranks is a few tens of
    milliseconds of NumPy copies): the thread that issues the collectives goes straight on to the next step."""
    global _parse_pool
    if _parse_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _parse_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="autoinst-gather")
    return _parse_pool


def _pack(local: dict):
    keys = sorted(local)
    lens = [int(local[k].shape[0]) for k in keys]
    # payload: [n_chunks, (chunk id, length) * n_chunks, labels...]
    head = np.array([len(keys)] + [v for kl in zip(keys, lens) for v in kl], dtype=np.int32)
    body = np.concatenate([np.asarray(local[k], dtype=np.int32) for k in keys]) if keys else np.zeros(0, np.int32)
    return np.concatenate([head, body])


def _unpack(a: np.ndarray, merged: dict):
    nc = int(a[0])
    off = 1 + 2 * nc
    for c in range(nc):
        k, ln = int(a[1 + 2 * c]), int(a[2 + 2 * c])
        merged[k] = a[off: off + ln].copy()
        off += ln


def gather_labels_async(local: dict, device=None, force: bool = False):
    """Rooted gather of ``{chunk_index: int32 label array}`` to rank 0 (counts first, then a direct
    send of every rank's arrays to the root -- on the GPUs one hop over xGMI with the ``nccl`` = RCCL backend, ``gloo`` in CPU
    tests).  Two collectives on the default process group: an ``all_reduce(MAX)`` of the payload length (8 bytes), then ONE rooted
    ``gather`` of the payloads padded to that length -- no rank but the root receives anything, and chunks of one size need no
    padding.  Must be called by every rank, by the same thread each time (collectives are ordered).

    Returns a ``concurrent.futures.Future`` on rank 0 whose result is the merged dict (the copies off the device and the
    parsing run on a helper thread, so the caller can issue the next step's exchange meanwhile) and ``None`` on the other
    ranks.  Without an initialised group (or with one rank, unless ``force``) the future holds ``dict(local)``.
    """
    from concurrent.futures import Future

    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        f = Future()
        f.set_result(dict(local))
        return f
    world, rank = dist.get_world_size(), dist.get_rank()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    payload = _pack(local)
    # (i) the longest payload, known to every rank (8 bytes); (ii) ONE rooted gather of the payloads, padded to that length: only
    # the root receives.  `gather` is the collective RCCL / NCCL implement as a group of direct sends to the root, which is the
    # exchange SURVEY section 5 specifies; chunks of one size (the bench) need no padding at all.
    size = torch.tensor([payload.shape[0]], dtype=torch.int64, device=device)
    dist.all_reduce(size, op=dist.ReduceOp.MAX)
    mx = int(size.item())
    buf = torch.zeros(mx, dtype=torch.int32, device=device)
    buf[: payload.shape[0]] = torch.from_numpy(payload).to(device)
    recv = [torch.empty(mx, dtype=torch.int32, device=device) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gather_list=recv, dst=0)
    if rank != 0:
        return None
    if device.type == "cuda":
        torch.cuda.current_stream(device).synchronize()   # the gather has landed before the helper thread reads it
    recv = {r: recv[r] for r in range(1, world)}

    def parse():
        merged = {}
        _unpack(payload, merged)
        for r in sorted(recv):
            _unpack(recv[r].cpu().numpy(), merged)
        return merged

    return _parser().submit(parse)


def gather_labels(local: dict, device=None, force: bool = False):
    """`gather_labels_async` and its result: the merged dict on rank 0, ``None`` elsewhere."""
    f = gather_labels_async(local, device=device, force=force)
    return None if f is None else f.result()


def run_chunks(chunks, *, threads: int = 2, batch: int = 12, device: int | None = None, alpha=None, theta=None, gamma=None,
               T=None, split_lim=None, contexts=None, builders: int = 1):
    """The chunk loop of ``run_pipeline.py:160-179`` for the chunks of ONE rank / GPU.

    ``chunks``: sequence of ``(points, tarl)`` or ``(points, tarl, dino)`` (host arrays or device tensors; ``tarl`` / ``dino``
    may be ``None``).  Returns the list of int32 label arrays (``label[i]`` = group of point ``i``, groups numbered in the
    reference's emission order), chunk by chunk -- what `ncuts_api.ncuts_labels` gives for each chunk alone.

    Chunks are batched largest first (``batch`` per call);
    ``threads`` host threads with one `Context` each (or the given ``contexts``) take batches from one queue, and the C calls
    release the GIL, so ``threads`` batched calls are in flight on the device together.  ``builders`` further threads (one
    `Context` each; 0: none) build the affinity graphs of the NEXT batches while the cuts run, at most ``threads`` batches ahead
    (`bench.py` measures this arrangement: +3 % over every thread building its own batch first).  A failure in any batch is raised
    after the other threads have finished their current batch.
    """
    import queue
    import threading

    from . import ncuts_api as api
    from .config import CONFIG, SPLIT_LIM

    cfg = {"alpha": CONFIG["alpha"] if alpha is None else alpha, "theta": CONFIG["theta"] if theta is None else theta,
           "gamma": CONFIG["gamma"] if gamma is None else gamma}
    T = CONFIG["T"] if T is None else T
    split_lim = SPLIT_LIM if split_lim is None else split_lim
    n_chunks = len(chunks)
    if n_chunks == 0:
        return []
    if threads <= 0 or batch <= 0:
        raise ValueError("threads and batch must be positive")
    sizes = [int(c[0].shape[0]) for c in chunks]
    order = sorted(range(n_chunks), key=lambda i: (-chunk_cost(sizes[i]), i))
    batches = [order[j:j + batch] for j in range(0, n_chunks, batch)]
    threads = min(threads, len(batches))
    own = contexts is None   # contexts made here are closed here; pass `contexts` to keep their workspaces between calls
    ctxs = list(contexts) if contexts is not None else [api.Context(device) for _ in range(threads)]
    if len(ctxs) < threads:
        raise ValueError("fewer contexts than threads")
    builders = max(0, min(int(builders), len(batches)))
    bctxs = [api.Context(device) for _ in range(builders)]
    work = queue.Queue()
    for b in batches:
        work.put(b)
    built = queue.Queue(maxsize=max(1, threads))   # (ids, graphs) ahead of the cuts; None = a builder has finished
    out = [None] * n_chunks
    errors = []

    def build(ids, ctx, graphs):
        for i in ids:
            c = chunks[i]
            graphs.append(api.build_affinity(c[0], c[1] if len(c) > 1 else None, c[2] if len(c) > 2 else None, ctx=ctx, **cfg))

    def builder(bi):
        while not errors:
            try:
                ids = work.get_nowait()
            except queue.Empty:
                return
            graphs = []
            try:
                build(ids, bctxs[bi], graphs)
            except BaseException as e:  # noqa: BLE001 -- re-raised in the caller's thread
                errors.append(e)
                for g in graphs:
                    g.free()
                return
            built.put((ids, graphs))

    def worker(w):
        while True:
            graphs = []
            try:
                if builders:
                    job = built.get()
                    if job is None:       # every builder has finished and the queue is drained (the caller's thread says so)
                        return
                    ids, graphs = job
                    if errors:
                        continue          # (drain: free what was built, cut nothing more)
                else:
                    if errors:
                        return
                    try:
                        ids = work.get_nowait()
                    except queue.Empty:
                        return
                    build(ids, ctxs[w], graphs)
                # the cut runs on THIS thread's context whichever context built the graphs (a context serves one thread at a time)
                if len(graphs) == 1:
                    lab, _, _ = api.ncuts_labels(graphs[0], sizes[ids[0]], T, split_lim, ctx=ctxs[w])
                    labs = [lab]
                else:
                    labs, _, _ = api.ncuts_labels_batch(graphs, [sizes[i] for i in ids], T, split_lim, ctx=ctxs[w])
                for i, lab in zip(ids, labs):
                    out[i] = lab
            except BaseException as e:  # noqa: BLE001 -- re-raised in the caller's thread
                errors.append(e)
            finally:
                for g in graphs:
                    g.free()

    ws = [threading.Thread(target=worker, args=(w,)) for w in range(threads)]
    bs = [threading.Thread(target=builder, args=(bi,)) for bi in range(builders)]
    for t in ws + bs:
        t.start()
    for t in bs:
        t.join()
    if builders:
        for _ in ws:
            built.put(None)   # behind everything the builders have queued
    for t in ws:
        t.join()
    for c in bctxs:
        c.close()
    if own:
        for c in ctxs:
            c.close()
    if errors:
        raise errors[0]
    return out
