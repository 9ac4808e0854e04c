#!/usr/bin/env python3
"""Diagnostic (stamps build): cycles and wall time of wave M's knot loop in k_backward4, stand-alone
(qilqr_backwards_pass) and as round 0 of a solve (max_iters = 1: the second round's backward kernel settles and
does not run the recursion, so the stamps left are round 0's).  cycles / time = the shader clock it ran at.
usage (from the repository root): PYTHONPATH=. python3 profiles/microbench/backward_clock.py"""
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
    o = out.astype(np.float64)
    cyc = o[:, [0, 1, 2, 4, 5, 6, 7]].sum(axis=1)
    us = (out[:, 3] & np.uint64(0xfffff)).astype(np.float64) / 100.0
    parts = [((out[:, 3] >> np.uint64(sh)) & np.uint64(0x7ff)).astype(np.float64) / 100.0 for sh in (20, 31, 42, 53)]
    pro = sum(parts)
    return (np.median(cyc), np.median(us), np.median(cyc / np.maximum(us, 1e-9)) / 1e3, np.median(pro), pro.max(),
            " + ".join("%.1f" % np.median(x) for x in parts))


cfg = pb.config2(B=B, N=N)
s = capi.from_config(cfg, force_general=4)
trajs = s.forward_sim(cfg["init"], np.zeros((B, N, 52)), 1.0)
for _ in range(3):
    s.backwards_pass(trajs)
print("stand-alone     : %.0f cycles, %.1f us, %.3f GHz; entry -> loop %.1f us (max %.1f) = loads %s: settle, barrier, ring fill" % stamps(s))
for mi in (1, 3, 8):
    cfg1 = dict(cfg, options=dict(cfg["options"], max_iters=mi, rtol=0.0, atol=0.0))
    s1 = capi.from_config(cfg1, force_general=4)
    for _ in range(3):
        s1.solve_batch(cfg["init"])
    print("in a solve, last full round of %d: %.0f cycles, %.1f us, %.3f GHz; entry -> loop %.1f us (max %.1f) = loads %s: settle, barrier, ring fill" % ((mi,) + stamps(s1)))
