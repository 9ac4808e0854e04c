#!/usr/bin/env python3
"""Diagnostic (round 4): ONE solver configuration in a process of its own (hardware queues go to streams in creation order, so
configurations compared inside one process do not see the same machine): whole device-resident solves of configs[1]'s problems
at batch size B.  usage: PYTHONPATH=. GPU_MAX_HW_QUEUES=8 python3 profiles/microbench/one_config.py B [key=value ...] [reps=5]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quadrotorilqr_amd import capi, problems as pb  # noqa: E402

B = int(sys.argv[1])
kw = dict(a.split("=") for a in sys.argv[2:])
reps = int(kw.pop("reps", 5))
N = int(kw.pop("N", 100))
seed = int(kw.pop("seed", 4))
kw = {k: (v if k == "precision" else int(v)) for k, v in kw.items()}
dev = torch.device("cuda", 0)
cfg = pb.config2(B=B, N=N, seed=seed)
init = torch.from_numpy(cfg["init"]).to(dev)
bufs = (torch.empty_like(init), torch.empty(B, dtype=torch.float64, device=dev), [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)])
s = capi.from_config(cfg, device=0, **kw)
for _ in range(2):
    s.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    s.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
t = float(np.median(ts))
print(f"B={B:6d} N={N} {' '.join(f'{k}={v}' for k, v in kw.items()):40s}: {t * 1e3:8.3f} ms (min {min(ts) * 1e3:.3f}) {B / t:9.0f} solves/s  moved {s.compaction_moves()}"
      f"  rounds<= {int(bufs[2][3].max())}", flush=True)
