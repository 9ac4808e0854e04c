#!/usr/bin/env python3
"""Diagnostic (round 4): where a knot of the fused backward wavefront (bw4_fused_wave, force_general = 5) spends its cycles, and
how often it had to wait for the loader's tag -- separate -DQILQR_STAMPS builds (make stamps; variants: libquadrotor_ilqr_<name>.so
built with -DQILQR_STAMPS plus its own defines).  Each stamp is an s_memtime + s_waitcnt lgkmcnt(0) (~70 cycles, and it closes the
LDS operations in flight), so the shares are of a slower loop than the product's.
usage (repository root): PYTHONPATH=. python3 profiles/microbench/fused_stamps.py [name ...]  (default: stamps)  B from env B=1"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quadrotorilqr_amd import problems as pb  # noqa: E402

names = sys.argv[1:] or ["stamps"]
N = 100
for B in [int(b) for b in os.environ.get("B", "1,4,1024").split(",")]:
    cfg = pb.config2(B=B, N=N)
    for name in names:
        sp = importlib.util.spec_from_file_location("capi_" + name, os.path.join(ROOT, "quadrotorilqr_amd", "capi.py"))
        m = importlib.util.module_from_spec(sp); sp.loader.exec_module(m)
        m.LIB_PATH = os.path.join(ROOT, "quadrotorilqr_amd", "lib", f"libquadrotor_ilqr_{name}.so")
        s = m.from_config(cfg, force_general=5, profile=3)
        tr = s.forward_sim(cfg["init"], np.zeros((B, N, 52)), 1.0)
        for _ in range(3):
            s.backwards_pass(tr)
        s.profile_reset()
        s.backwards_pass(tr)
        p = s.profile_get()
        out = np.zeros((B, 8), dtype=np.uint64)
        m.load().qilqr_debug_stamps(s._h, out.ctypes.data_as(C.c_void_p), C.c_int32(B))
        med = np.median(out.astype(np.float64), axis=0) / N
        lab = ["T, H (6 MFMA), M^T V_x", "tag check, operand reads", "gather, Q_u, rhs", "4x4 solve", "stores, Q_u^T k, V_x, shuffles", "V_xx MFMA, post"]
        if "pipe" in name:  # bw4_fused_wave_pipelined's sections
            lab = ["T (3 MFMA) + prev. V_x, shuffles, Q_u^T k, stores", "H (3 MFMA) + M^T V_x, butterflies", "tag check, operand reads, post",
                   "gather, Q_uu / Q_u broadcasts, rhs", "LDL^T + solve", "operand select, V_xx MFMA"]
        tot = med[:6].sum()
        print(f"{name} B={B}: launch {p['backward_ms'] * 1e3 / max(p['backward_launches'], 1):.1f} us; {tot:.0f} stamped cycles per knot; "
              f"knots that waited for the tag: {med[6] * N:.0f} of {N}, spins per knot {med[7]:.2f}")
        for n_, v in zip(lab, med[:6]):
            print(f"    {n_:34s} {v:7.0f} cycles/knot {100 * v / tot:5.1f} %")
        s.close()
