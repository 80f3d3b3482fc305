import torch, time
dev = torch.device("cuda", 0)
h = torch.empty(159 * 1024 * 1024 // 8, dtype=torch.float64).pin_memory()
d = torch.empty_like(h, device=dev)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(6):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("H2D pinned 6 x 159 MiB: %.1f ms = %.1f GB/s" % (dt * 1e3, 6 * h.numel() * 8 / dt / 1e9), flush=True)
