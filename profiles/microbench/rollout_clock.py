#!/usr/bin/env python3
"""Diagnostic (stamps build): wall time of wave X's knot loop in k_rollout3 and of what precedes it, stand-alone
(qilqr_forward_sim) and as the last full round of a solve.
usage (from the repository root): PYTHONPATH=. python3 profiles/microbench/rollout_clock.py"""
import ctypes as C
import os

import numpy as np

from quadrotorilqr_amd import capi, problems as pb

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi.LIB_PATH = os.path.join(ROOT, "quadrotorilqr_amd", "lib", "libquadrotor_ilqr_stamps.so")
B, N = 1024, 100


def stamps(s):
    out = np.zeros((B, 8), dtype=np.uint64)
    capi.load().qilqr_debug_stamps(s._h, out.ctypes.data_as(C.c_void_p), C.c_int32(B))
    st = out.reshape(-1)[: (B // 64) * 24].reshape(B // 64, 3, 8)
    x = st[:, 0, :]
    cyc = x[:, [0, 1, 2, 4, 5, 6]].astype(np.float64).sum(axis=1)
    loop = (x[:, 3] & np.uint64(0xfffff)).astype(np.float64) / 100.0
    pro = ((x[:, 3] >> np.uint64(20)) & np.uint64(0xfffff)).astype(np.float64) / 100.0
    return np.median(cyc), np.median(loop), np.median(cyc / loop) / 1e3, np.median(pro), pro.max()


cfg = pb.config2(B=B, N=N)
s = capi.from_config(cfg)
conv = s.solve_batch(cfg["init"])["traj"]
gains, _ = s.backwards_pass(conv)
for _ in range(3):
    s.forward_sim(conv, gains, 1.0)
print("stand-alone, next to the nominal trajectory: %.0f cycles, loop %.1f us, %.3f GHz; entry -> loop %.1f us (max %.1f)" % stamps(s))
for mi in (1, 3, 8, 20):
    cfg1 = dict(cfg, options=dict(cfg["options"], max_iters=mi, rtol=0.0, atol=0.0))
    s1 = capi.from_config(cfg1)
    for _ in range(3):
        s1.solve_batch(cfg["init"])
    print("in a solve, last rollout of max_iters = %2d: %.0f cycles, loop %.1f us, %.3f GHz; entry -> loop %.1f us (max %.1f)" % ((mi,) + stamps(s1)))
