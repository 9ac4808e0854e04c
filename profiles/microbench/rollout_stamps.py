#!/usr/bin/env python3
"""Diagnostic: per-knot cycle shares of the three roles of k_rollout3 (separate -DQILQR_STAMPS build)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quadrotorilqr_amd import capi, problems as pb  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "quadrotorilqr_amd", "lib", "libquadrotor_ilqr_stamps.so")
B, N = 1024, 100
cfg = pb.config2(B=B, N=N)
s = capi.from_config(cfg, single_wave_rollout=2)  # the stamps live in k_rollout3
mode = sys.argv[1] if len(sys.argv) > 1 else "converged"
if mode == "first":   # first iteration: the rollout is far from the nominal trajectory (general branches)
    trajs = s.forward_sim(cfg["init"], np.zeros((B, N, 52)), 1.0)
else:                 # late iteration: rollout next to the nominal trajectory (series branches)
    trajs = s.solve_batch(cfg["init"])["traj"]
gains, _ = s.backwards_pass(trajs)
for _ in range(3):
    s.forward_sim(trajs, gains, 1.0)
print("mode:", mode)
out = np.zeros((B, 8), dtype=np.uint64)
capi.load().qilqr_debug_stamps(s._h, out.ctypes.data_as(C.c_void_p), C.c_int32(B))
blocks = B // 64
st = out.reshape(-1)[: blocks * 24].reshape(blocks, 3, 8).astype(np.float64)
names = {0: ["operands from LDS", "rho, dx", "control law", "-", "stores, acceleration, v, LDS write", "barrier wait", "LDS read", "-"],
         1: ["nominal pose from LDS", "T <- T Exp(dt v)", "pose part of (-)", "-", "LDS write + stores", "barrier wait", "LDS read", "-"],
         2: ["load issue", "wait for loads, LDS writes", "-", "-", "-", "barrier wait", "-", "-"]}
for role in (0, 1, 2):
    med = np.median(st[:, role, :], axis=0) / N
    print("wave", ["X (control)", "Y (pose)", "L (loader)"][role], " total %.0f cycles/knot" % med.sum())
    for n_, m in zip(names[role], med):
        if m > 0:
            print(f"   {n_:34s} {m:8.0f}  {100 * m / med.sum():5.1f} %")
