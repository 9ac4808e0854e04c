#!/usr/bin/env python3
"""Diagnostic: whole-solve time of the persistent solve (k_solve4) against the rounds, by batch size (device-resident)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quadrotorilqr_amd import capi, problems as pb
dev = torch.device("cuda", 0)
for B, seed in [(256, 2), (1024, 2), (1536, 2), (2048, 2), (3072, 2), (4096, 4), (8192, 4)]:
    cfg = pb.config2(B=B, N=100, seed=seed)
    init = torch.from_numpy(cfg["init"]).to(dev)
    res = {}
    for name, pp, rk in (("rounds", 2, 0), ("persistent", 1, 0), ("rounds16", 2, 3), ("rounds3", 2, 2)):
        s = capi.from_config(cfg, persistent=pp, single_wave_rollout=rk)
        bufs = (torch.empty_like(init), torch.empty(B, dtype=torch.float64, device=dev), [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)])
        for _ in range(3):
            s.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            s.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / reps
        s.close()
    print("B %6d  rounds %8.3f ms (%7.0f solves/s)   persistent %8.3f ms (%7.0f solves/s)   ratio %.2f   rounds with k_rollout16 %8.3f ms, with k_rollout3 %8.3f ms" %
          (B, res["rounds"] * 1e3, B / res["rounds"], res["persistent"] * 1e3, B / res["persistent"], res["rounds"] / res["persistent"],
           res["rounds16"] * 1e3, res["rounds3"] * 1e3), flush=True)
