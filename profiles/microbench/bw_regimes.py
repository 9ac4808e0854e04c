#!/usr/bin/env python3
"""Diagnostic (stamps build): wave M of k_backward4 by how many trajectories share the chip -- per-section cycles per knot
(median over trajectories), the knot loop's wall time on the constant 100 MHz clock, hence the shader clock it ran at.
usage (repository root, after `make -C quadrotorilqr_amd/csrc stamps`): PYTHONPATH=. python3 profiles/microbench/bw_regimes.py"""
import ctypes as C
import os

import numpy as np

from quadrotorilqr_amd import capi, problems as pb

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi.LIB_PATH = os.path.join(ROOT, "quadrotorilqr_amd", "lib", "libquadrotor_ilqr_stamps.so")
N = 100
names = ["reads+T", "H", "gather+bcast", "-", "LDLT+solve", "stores+handoff", "Vxx MFMA", "barrier"]
for B in (4, 64, 256, 512, 1024, 2048):
    cfg = pb.config2(B=B, N=N)
    s = capi.from_config(cfg, force_general=4)
    trajs = s.forward_sim(cfg["init"], np.zeros((B, N, 52)), 1.0)
    for _ in range(3):
        s.backwards_pass(trajs)
    out = np.zeros((B, 8), dtype=np.uint64)
    capi.load().qilqr_debug_stamps(s._h, out.ctypes.data_as(C.c_void_p), C.c_int32(B))
    o = out.astype(np.float64)
    us = (out[:, 3] & np.uint64(0xfffff)).astype(np.float64) / 100.0
    sec = o.copy()
    sec[:, 3] = 0
    cyc = sec.sum(axis=1)
    med = np.median(sec, axis=0) / N
    # by the wave's place in its block (trajectory index mod 4): M0 shares its SIMD with G, M1 with L
    byw = [np.median(cyc[w::4]) / N for w in range(4)] if B >= 4 else []
    print(f"B={B:5d}: {np.median(cyc)/N:7.0f} cycles/knot, loop {np.median(us):6.1f} us (max {us.max():6.1f}), clock {np.median(cyc/np.maximum(us,1e-9))/1e3:5.3f} GHz | "
          + " ".join(f"{n}={m:.0f}" for n, m in zip(names, med) if n != "-") + " | by wave " + " ".join(f"{x:.0f}" for x in byw))
    s.close()
