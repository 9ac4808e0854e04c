#!/usr/bin/env python3
"""Diagnostic (round 4): are the first rounds of configs[1] slow because every trajectory is live, or because the first iterates are far
from the hover pose (rotation errors beyond the series' ranges: closed forms with sqrt / atan2 / sincos)?  The same batch with initial
rotation errors of pi/4 (configs[1]) and of 0.2 rad, for a kernel trace (rounds_per_launch = 1: one round per launch).
usage: rocprofv3 --kernel-trace ... -- python3 profiles/microbench/angle_probe.py <ang_rad>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quadrotorilqr_amd import capi, problems as pb  # noqa: E402

ang = float(sys.argv[1]) if len(sys.argv) > 1 else np.pi / 4
dev = torch.device("cuda", 0)
cfg = pb.config2(B=1024, N=100, seed=2)
cfg["init"] = pb.random_start_batch(np.arange(1024), cfg["desired"], 2, ang_rad=ang)
init = torch.from_numpy(cfg["init"]).to(dev)
B = 1024
bufs = (torch.empty_like(init), torch.empty(B, dtype=torch.float64, device=dev), [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)])
s = capi.from_config(cfg, device=0, rounds_per_launch=1)
for _ in range(3):
    s.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
torch.cuda.synchronize()
print("ang", ang, "iters mean", float(bufs[2][1].float().mean()), "max", int(bufs[2][1].max()))
