#!/usr/bin/env python3
"""Diagnostic: the part of a batch solve that is not rounds.  max_iters = 0 runs begin + retile + linearise +
init + gather + the final synchronisation only; max_iters = k adds k rounds.
usage (from the repository root): PYTHONPATH=. python3 profiles/microbench/fixed_cost.py"""
import time
import torch
from quadrotorilqr_amd import capi, problems as pb
dev = torch.device("cuda:0")
B, N = 1024, 100
for mi in (0, 1, 2, 3, 5, 9):
    cfg = pb.config2(B=B, N=N)
    cfg["options"] = dict(cfg["options"], max_iters=mi)
    s = capi.from_config(cfg)
    init = torch.tensor(cfg["init"], device=dev)
    out = torch.empty_like(init)
    cost = torch.empty(B, dtype=torch.float64, device=dev)
    ints = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)]
    for _ in range(5):
        s.solve_batch_device(init, out, cost, *ints)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 50
    for _ in range(K):
        s.solve_batch_device(init, out, cost, *ints)
    torch.cuda.synchronize()
    print(f"max_iters {mi}: {(time.perf_counter() - t0) / K * 1e6:.1f} us per solve, n_fwd max {int(ints[3].max())}", flush=True)
    s.close()
