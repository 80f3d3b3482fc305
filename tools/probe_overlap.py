"""Does a pinned-host -> device copy on its own stream slow down a bandwidth-bound kernel stream?  python tools/probe_overlap.py"""
import time, json, torch
dev = torch.device("cuda", 0)
a = torch.empty(1 << 27, dtype=torch.float64, device=dev).fill_(1.0)   # 1 GiB
b = torch.empty_like(a)
h = torch.empty(240_000_000, dtype=torch.float64).pin_memory()          # 1.92 GB: one batch of twelve chunks' inputs
d = torch.empty_like(h, device=dev)
cs = torch.cuda.Stream(device=dev)
def kernels(n):
    for _ in range(n):
        b.copy_(a)
def run(with_copy, n=100):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    copies = 0
    if with_copy:
        with torch.cuda.stream(cs):
            for _ in range(3):
                d.copy_(h, non_blocking=True); copies += 1
    kernels(n)
    torch.cuda.current_stream().synchronize()
    t_k = time.perf_counter() - t0
    cs.synchronize()
    t_all = time.perf_counter() - t0
    return t_k, t_all, copies
run(False, 10); run(True, 10)
k0, _, _ = run(False)
k1, all1, c = run(True)
print(json.dumps({"kernel_loop_alone_ms": 1e3 * k0, "kernel_loop_beside_copies_ms": 1e3 * k1, "slowdown": k1 / k0,
                  "copies": c, "copy_GBps_beside_kernels": c * h.numel() * 8 / all1 / 1e9,
                  "d2d_GBps_alone": 100 * 2 * a.numel() * 8 / k0 / 1e9}))
