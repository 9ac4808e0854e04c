#!/usr/bin/env python3
"""Diagnostic: whole-solve rate for every pair (backward kernel, rollout kernel) at large batches -- where the automatic
choices (ilqr_capi.hip: backward_kind, launch_rollout) should change.
usage (from the repository root): PYTHONPATH=. python3 profiles/microbench/kernel_grid.py [B ...]"""
import sys, time
import torch
from quadrotorilqr_amd import capi, problems as pb
dev = torch.device("cuda:0")
BWD = ((0, "auto"), (4, "bw4"), (3, "bw2"), (2, "bw1"))
ROL = ((0, "auto"), (3, "r16"), (2, "r3"), (1, "r1"))
for B in [int(x) for x in sys.argv[1:]] or [4096, 8192, 16384, 65536]:
    cfg = pb.config2(B=B, N=100, seed=4)
    init = torch.from_numpy(cfg["init"]).to(dev)
    out = torch.empty_like(init); cost = torch.empty(B, dtype=torch.float64, device=dev)
    ints = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)]
    for kb, nb in BWD:
        line = []
        for kr, nr in ROL:
            if (kb == 0) != (kr == 0):
                continue
            s = capi.from_config(cfg, force_general=kb, single_wave_rollout=kr)
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t = time.perf_counter()
                s.solve_batch_device(init, out, cost, *ints)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
            line.append(f"{nb}+{nr} {B / best / 1e3:.1f}k")
            s.close()
        print(f"B={B}: " + "  ".join(line), flush=True)
