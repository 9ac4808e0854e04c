#!/usr/bin/env python3
"""Diagnostic (round 6): the host-buffer batch solve (qilqr_solve_batch, B = 1024, pinned buffers) with the late finishers written straight
into the caller's arrays against the staged form (compact block + host scatter; forced through the diagnostics build), with and without
four more streams created first in the process (hardware queues go to streams in creation order: bench.py's full run creates the B = 8192
solver's sub-batch streams before this leg).  usage (repository root): GPU_MAX_HW_QUEUES=8 PYTHONPATH=. python3 profiles/microbench/h2h_ab.py [extra_streams]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
sys.path.insert(0, "tests")
from diag_lib import capi_diag
from quadrotorilqr_amd import problems as pb
d = capi_diag()
extra = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = pb.config2(B=1024, N=100, seed=2)
dev = torch.device("cuda:0")
keep = []
if extra:  # a B = 8192 handle's sub-batch streams, created by one untimed solve
    big = pb.config2(B=8192, N=100, seed=4)
    ls = d.from_config(big)
    li = torch.from_numpy(big["init"]).to(dev)
    lb = (torch.empty_like(li), torch.empty(8192, dtype=torch.float64, device=dev), [torch.empty(8192, dtype=torch.int32, device=dev) for _ in range(4)])
    ls.solve_batch_device(li, lb[0], lb[1], *lb[2]); torch.cuda.synchronize(); keep.append(ls)
s = d.from_config(cfg)
hin = d.host_array(cfg["init"].shape); hin[...] = cfg["init"]
hout = dict(traj=d.host_array(cfg["init"].shape), cost=d.host_array((1024,)), **{k: d.host_array((1024,), np.int32) for k in ("status", "iters", "n_bwd", "n_fwd")})
init = torch.from_numpy(cfg["init"]).to(dev)
ob = (torch.empty_like(init), torch.empty(1024, dtype=torch.float64, device=dev), [torch.empty(1024, dtype=torch.int32, device=dev) for _ in range(4)])
sd = d.from_config(cfg)
for _ in range(30):
    s.solve_batch(hin, out=hout)
res = {"direct": [], "staged": [], "device": []}
for rep in range(40):
    for name in ("direct", "staged"):
        d.load().qilqr_debug_set_staged_late(1 if name == "staged" else 0)
        t = time.perf_counter(); s.solve_batch(hin, out=hout); res[name].append(time.perf_counter() - t)
    t = time.perf_counter(); sd.solve_batch_device(init, ob[0], ob[1], *ob[2]); res["device"].append(time.perf_counter() - t)
d.load().qilqr_debug_set_staged_late(0)
m = {k: float(np.median(v)) * 1e3 for k, v in res.items()}
print(f"extra streams first: {extra}: device-resident {m['device']:.3f} ms; pinned host buffers: direct late part {m['direct']:.3f} (+{m['direct'] - m['device']:.3f}), staged {m['staged']:.3f} (+{m['staged'] - m['device']:.3f})")
