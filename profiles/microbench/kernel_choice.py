#!/usr/bin/env python3
"""Diagnostic: whole-solve rate per backward kernel at the batch sizes where the automatic choice changes.
usage (from the repository root): PYTHONPATH=. python3 profiles/microbench/kernel_choice.py [B ...]"""
import sys, time
import torch
from quadrotorilqr_amd import capi, problems as pb
dev = torch.device("cuda:0")
for B in [int(x) for x in sys.argv[1:]] or [512, 768, 2048, 4096, 8192]:
    cfg = pb.config2(B=B, N=100, seed=4)
    init = torch.from_numpy(cfg["init"]).to(dev)
    out = torch.empty_like(init); cost = torch.empty(B, dtype=torch.float64, device=dev)
    ints = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)]
    line = []
    for kern, name in ((0, "auto"), (4, "k_backward4"), (3, "k_backward2"), (2, "one wave")):
        s = capi.from_config(cfg, force_general=kern)
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t = time.perf_counter()
            s.solve_batch_device(init, out, cost, *ints)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        line.append(f"{name} {B / best / 1e3:.1f}k")
        s.close()
    print(f"B={B}: " + "  ".join(line), flush=True)
