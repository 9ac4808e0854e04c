#!/usr/bin/env python3
"""Diagnostic (round 4): what compaction of the live trajectories could buy in the saturated regime.  The problems of a batch are
solved once, then again in the order of their round counts (longest first): the live ones are then a dense prefix of the batch in
every round, which is what a free and perfect compaction would give.  Whole device-resident solves.
usage (repository root): PYTHONPATH=. GPU_MAX_HW_QUEUES=8 python3 profiles/microbench/sorted_batch.py [B ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quadrotorilqr_amd import capi, problems as pb  # noqa: E402

dev = torch.device("cuda", 0)


def timed(s, init, bufs, reps=3):
    for _ in range(2):
        s.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        s.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for B in [int(a) for a in sys.argv[1:]] or [8192]:
    cfg = pb.config2(B=B, N=100, seed=4)
    init = torch.from_numpy(cfg["init"]).to(dev)
    bufs = (torch.empty_like(init), torch.empty(B, dtype=torch.float64, device=dev), [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)])
    for label, kw in [("default", {}), ("never compacting", dict(compaction=-1)), ("one stream", dict(streams=1)),
                      ("one stream, never", dict(streams=1, compaction=-1))]:
        s = capi.from_config(cfg, device=0, **kw)
        t = timed(s, init, bufs)
        rounds = (bufs[2][3]).cpu().numpy().astype(np.int64)  # n_fwd: one rollout per round
        cost = bufs[1].cpu().numpy().copy()
        order = np.argsort(-rounds, kind="stable")
        init_sorted = init[torch.from_numpy(order).to(dev)].contiguous()
        ts = timed(s, init_sorted, bufs)
        same = np.array_equal(bufs[1].cpu().numpy(), cost[order])
        print(f"B={B:6d} {label:18s}: as given {t * 1e3:8.2f} ms {B / t:9.0f} solves/s | longest first {ts * 1e3:8.2f} ms {B / ts:9.0f} solves/s "
              f"(x{t / ts:.3f}; same costs: {same})", flush=True)
        s.close()
