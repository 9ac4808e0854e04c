#!/usr/bin/env python3
"""Diagnostic: what one solve of configs[1] (B = 1024, N = 100, fp64, device-resident) is made of -- average launch time and
count of every kernel class with every launch timed (profile = 2: the timing itself costs a few per cent), beside the
untimed wall time of the same solve.  usage (from the repository root): python profiles/microbench/round_breakdown.py [B]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quadrotorilqr_amd import capi, problems as pb
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = pb.config2(B=B, N=100, seed=2)
dev = torch.device("cuda", 0)
init = torch.from_numpy(cfg["init"]).to(dev)
bufs = (torch.empty_like(init), torch.empty(B, dtype=torch.float64, device=dev), [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)])
for prof in (0, 2):
    s = capi.from_config(cfg, profile=prof)
    for _ in range(3):
        s.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
    torch.cuda.synchronize()
    if prof:
        s.profile_reset()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        s.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    print("profile %d: %.3f ms per solve (%.0f solves/s)" % (prof, wall * 1e3, B / wall))
    if prof:
        p = s.profile_get()
        tot = 0.0
        for k in ("backward", "rollout", "linearize", "other"):
            n = p[k + "_launches"]
            tot += p[k + "_ms"] / reps
            print("   %-10s %5.1f launches per solve, %7.2f us each, %7.3f ms per solve" % (k, n / reps, 1e3 * p[k + "_ms"] / max(n, 1), p[k + "_ms"] / reps))
        print("   kernels %.3f ms of %.3f ms per solve" % (tot, wall * 1e3))
