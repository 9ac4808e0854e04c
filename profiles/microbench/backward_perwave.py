#!/usr/bin/env python3
"""Diagnostic (stamps build): k_backward4's knot and barrier cycles per matrix wave index w = b mod 4.  Wave 4
(G) shares SIMD 0 with matrix wave 0.  usage: PYTHONPATH=. python3 profiles/microbench/backward_perwave.py"""
import ctypes as C
import os
import numpy as np
from quadrotorilqr_amd import capi, problems as pb
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi.LIB_PATH = os.path.join(ROOT, "quadrotorilqr_amd", "lib", "libquadrotor_ilqr_stamps.so")
B, N = 1024, 100
cfg = pb.config2(B=B, N=N)
s = capi.from_config(cfg, force_general=4)
trajs = s.forward_sim(cfg["init"], np.zeros((B, N, 52)), 1.0)
for _ in range(3):
    s.backwards_pass(trajs)
out = np.zeros((B, 8), dtype=np.uint64)
capi.load().qilqr_debug_stamps(s._h, out.ctypes.data_as(C.c_void_p), C.c_int32(B))
o = out.astype(np.float64)
for w in range(4):
    sel = o[w::4]
    print(f"matrix wave {w}: knot {np.median(sel[:, 6]) / N:7.0f} cycles, barrier {np.median(sel[:, 7]) / N:6.0f}")
