#!/usr/bin/env python3
"""Diagnostic: pinned host <-> device copy time of one configs[1] batch (14.7 MB) on this box, hipMemcpyAsync through torch."""
import time
import torch
n = 1024 * 100 * 18
h = torch.empty(n, dtype=torch.float64).pin_memory()
d = torch.empty(n, dtype=torch.float64, device="cuda")
for name, fn in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{name} {n * 8 / 1e6:.1f} MB pinned: median {ts[10] * 1e3:.3f} ms = {n * 8 / ts[10] / 1e9:.1f} GB/s (min {ts[0] * 1e3:.3f} ms)")
