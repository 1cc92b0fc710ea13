#!/usr/bin/env python3
"""Device copy bandwidth against buffer size (what a pure streaming kernel reaches beside / beyond the Infinity Cache)."""
import time
import torch

torch.cuda.set_device(0)
for mb in (64, 128, 256, 512, 1024, 2048, 4096):
    n = mb * 1024 * 1024 // 8
    a = torch.empty(n, dtype=torch.float64, device="cuda").normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    reps = max(10, 20000 // mb)
    t0 = time.perf_counter()
    for _ in range(reps):
        b.copy_(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"copy {mb:5d} MB -> {mb:5d} MB: {1e6 * dt:9.1f} us, {2 * mb * 1.048576e6 / dt / 1e9:7.0f} GB/s read + write", flush=True)
    del a, b
