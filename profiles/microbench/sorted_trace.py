#!/usr/bin/env python3
"""Diagnostic (round 4): the solves sorted_batch.py times, for a kernel trace: B problems on ONE stream, twice as given, then twice
longest-first.  usage: rocprofv3 --kernel-trace ... -- python3 profiles/microbench/sorted_trace.py [B] [streams]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quadrotorilqr_amd import capi, problems as pb  # noqa: E402

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = pb.config2(B=B, N=100, seed=4)
init = torch.from_numpy(cfg["init"]).to(dev)
bufs = (torch.empty_like(init), torch.empty(B, dtype=torch.float64, device=dev), [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)])
s = capi.from_config(cfg, device=0, streams=streams)
for _ in range(2):
    s.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
torch.cuda.synchronize()
rounds = bufs[2][3].cpu().numpy().astype(np.int64)
order = np.argsort(-rounds, kind="stable")
print("live per round:", [int((rounds > k).sum()) for k in range(int(rounds.max()))])
init_sorted = init[torch.from_numpy(order).to(dev)].contiguous()
for _ in range(2):
    s.solve_batch_device(init_sorted, bufs[0], bufs[1], *bufs[2])
torch.cuda.synchronize()
s.close()
