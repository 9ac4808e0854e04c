#!/usr/bin/env python3
"""Diagnostic: whole-solve rate by the number of sub-batches on their own streams (qilqr_device_config.streams) at large
batches -- where auto_parts (ilqr_capi.hip) should put its thresholds.
usage (from the repository root): PYTHONPATH=. python3 profiles/microbench/streams_sweep.py [B ...]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from quadrotorilqr_amd import capi, problems as pb
dev = torch.device("cuda:0")
for B in [int(x) for x in sys.argv[1:]] or [4096, 8192, 16384, 65536]:
    cfg = pb.config2(B=B, N=100, seed=4)
    init = torch.from_numpy(cfg["init"]).to(dev)
    out = torch.empty_like(init); cost = torch.empty(B, dtype=torch.float64, device=dev)
    ints = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)]
    line = []
    for k in (0, 1, 2, 3, 4, 6, 8):
        s = capi.from_config(cfg, streams=k)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            s.solve_batch_device(init, out, cost, *ints)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        line.append(f"{'auto' if k == 0 else k}: {B / best / 1e3:.1f}k")
        s.close()
    print(f"B={B}  streams " + "  ".join(line), flush=True)
