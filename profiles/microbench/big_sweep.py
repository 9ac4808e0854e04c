#!/usr/bin/env python3
"""Diagnostic (round 4): the saturated regime (B = 8192, the shard one GPU solves in configs[3]; and 65536) with the kernel
families forced: whole device-resident solves, ms and solves/s, one fresh process per line would be cleaner (hardware queues go
to streams in creation order) -- here one process, each configuration on its own handle, three repeats after a warm-up.
usage (repository root): PYTHONPATH=. GPU_MAX_HW_QUEUES=8 python3 profiles/microbench/big_sweep.py [B ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quadrotorilqr_amd import capi, problems as pb  # noqa: E402

dev = torch.device("cuda", 0)
for B in [int(a) for a in sys.argv[1:]] or [8192]:
    cfg = pb.config2(B=B, N=100, seed=4)
    init = torch.from_numpy(cfg["init"]).to(dev)
    bufs = (torch.empty_like(init), torch.empty(B, dtype=torch.float64, device=dev), [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)])
    for label, kw in [("default", {}), ("fused backward (5)", dict(force_general=5)), ("k_rollout16 (3)", dict(single_wave_rollout=3)),
                      ("fused + k_rollout16", dict(force_general=5, single_wave_rollout=3)), ("default, 2 streams", dict(streams=2)),
                      ("fused, 2 streams", dict(force_general=5, streams=2)), ("fused, 8 streams", dict(force_general=5, streams=8)),
                      ("fused + r16, 8 streams", dict(force_general=5, single_wave_rollout=3, streams=8))]:
        s = capi.from_config(cfg, device=0, **kw)
        for _ in range(2):
            s.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            s.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 3
        print(f"B={B:6d} {label:26s}: {t * 1e3:8.2f} ms  {B / t:10.0f} solves/s", flush=True)
        s.close()
