#!/usr/bin/env python3
"""Diagnostic: whole-solve time by how many rounds the host keeps the stream ahead of the device
(qilqr_device_config.sync_every).  usage (from the repository root): python profiles/microbench/sync_sweep.py [B ...]"""
import sys, time
import torch
sys.path.insert(0, ".")
from quadrotorilqr_amd import capi, problems as pb
dev = torch.device("cuda", 0)
for B in [int(x) for x in sys.argv[1:]] or [1024, 8192]:
    cfg = pb.config2(B=B, N=100, seed=2)
    init = torch.from_numpy(cfg["init"]).to(dev)
    out = torch.empty_like(init); cost = torch.empty(B, dtype=torch.float64, device=dev)
    ints = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)]
    line = []
    for k in (1, 2, 3, 4, 6):
        s = capi.from_config(cfg, sync_every=k)
        for _ in range(3):
            s.solve_batch_device(init, out, cost, *ints)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            s.solve_batch_device(init, out, cost, *ints)
        torch.cuda.synchronize()
        line.append(f"{k}: {(time.perf_counter() - t0) * 100:.3f} ms")
        s.close()
    print(f"B={B}  sync_every " + "  ".join(line), flush=True)
