#!/usr/bin/env python3
"""Diagnostic: k_backward4's launch time by how many matrix waves of a block are live (B = 1..4: one block; the
production library, HIP events on the dispatch) and, with the stamps build, the cycles per knot of each live wave.
usage (repository root): PYTHONPATH=. python3 profiles/microbench/bw_live.py"""
import ctypes as C
import os

import numpy as np

from quadrotorilqr_amd import capi, problems as pb

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = 100
for B in (1, 2, 3, 4, 8, 1021, 1024):
    cfg = pb.config2(B=B, N=N)
    s = capi.from_config(cfg, force_general=4, profile=3)
    trajs = s.forward_sim(cfg["init"], np.zeros((B, N, 52)), 1.0)
    for _ in range(3):
        s.backwards_pass(trajs)
    s.profile_reset()
    for _ in range(20):
        s.backwards_pass(trajs)
    p = s.profile_get()
    print(f"B={B:5d}: k_backward4 {p['backward_ms'] * 1e3 / max(p['backward_launches'], 1):7.2f} us per launch ({p['backward_launches']} launches)")
    s.close()
if os.path.exists(os.path.join(ROOT, "quadrotorilqr_amd", "lib", "libquadrotor_ilqr_stamps.so")):
    capi._lib = None
    capi.LIB_PATH = os.path.join(ROOT, "quadrotorilqr_amd", "lib", "libquadrotor_ilqr_stamps.so")
    for B in (1, 2, 3, 4):
        cfg = pb.config2(B=B, N=N)
        s = capi.from_config(cfg, force_general=4)
        trajs = s.forward_sim(cfg["init"], np.zeros((B, N, 52)), 1.0)
        for _ in range(3):
            s.backwards_pass(trajs)
        out = np.zeros((B, 8), dtype=np.uint64)
        capi.load().qilqr_debug_stamps(s._h, out.ctypes.data_as(C.c_void_p), C.c_int32(B))
        o = out.astype(np.float64)
        o[:, 3] = 0
        print(f"stamps B={B}: cycles/knot per wave", (o.sum(axis=1) / N).round(0), "sections of wave 0", (o[0] / N).round(0))
        s.close()
