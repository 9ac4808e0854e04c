#!/usr/bin/env python3
"""Diagnostic: two builds of the library on the same box, alternately, on large batches (boards differ by a few
per cent on the bandwidth-bound configurations, so an A/B across gpurun calls is not one).
usage (from the repository root): PYTHONPATH=. python3 profiles/microbench/ab_big.py libA.so libB.so [libC.so ...] [B ...]"""
import sys, time
import torch
from quadrotorilqr_amd import capi, problems as pb
libs = [a for a in sys.argv[1:] if a.endswith('.so')]
sizes = [int(a) for a in sys.argv[1:] if not a.endswith('.so')] or [65536]
dev = torch.device("cuda:0")
for B in sizes:
    cfg = pb.config2(B=B, N=100, seed=4)
    init = torch.from_numpy(cfg["init"]).to(dev)
    out = torch.empty_like(init); cost = torch.empty(B, dtype=torch.float64, device=dev)
    ints = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)]
    for rep in range(2):
        for lib in libs:
            capi.LIB_PATH = lib; capi._lib = None
            s = capi.from_config(cfg)
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t = time.perf_counter()
                s.solve_batch_device(init, out, cost, *ints)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
            print(f"B={B} {lib.split('/')[-1]:28s} {B / best:10.0f} solves/s", flush=True)
            s.close()
