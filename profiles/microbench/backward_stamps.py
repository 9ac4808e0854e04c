#!/usr/bin/env python3
"""Diagnostic: where a k_backward loop iteration spends its cycles (separate -DQILQR_STAMPS build).
Prints the per-knot cycle share of each section, median over trajectories."""
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
s = capi.from_config(cfg, force_general={"1": 2, "2": 3, "4": 4, "g": 1}[os.environ.get("BW", "4")])
trajs = s.forward_sim(cfg["init"], np.zeros((B, N, 52)), 1.0)
for _ in range(3):
    s.backwards_pass(trajs)
out = np.zeros((B, 8), dtype=np.uint64)
capi.load().qilqr_debug_stamps(s._h, out.ctypes.data_as(C.c_void_p), C.c_int32(B))
if os.environ.get("BW", "4") == "4":
    out[:, 3] = 0  # k_backward4 keeps wall-clock stamps there (backward_clock.py)
med = np.median(out.astype(np.float64), axis=0) / N
if os.environ.get("BW", "4") == "g":   # the general kernel k_backward<false> (force_general = 1: the reference's forms)
    names = ["prefetch issue (7 loads)", "T = V M (3 MFMA)", "H = C + M^T T (3 MFMA)", "gradient (3 FMA + 2 shuffles)",
             "H to LDS, barrier, Quu/Qu/rhs reads", "pivoted LDLT + solve", "K^T Quu, V_x, terms", "V_xx MFMA, stores, transpose via LDS"]
elif os.environ.get("BW", "4") == "1":   # one-wavefront kernel (force_general = 2)
    names = ["prefetch issue (7 loads)", "T = V M (3 MFMA)", "H = C + M^T T (3 MFMA)", "gradient (3 FMA + 2 shuffles)",
             "Quu/Qu/rhs broadcast", "LDLT + two solves", "V_x, terms", "V_xx MFMA, stores, hand-off"]
elif os.environ.get("BW", "4") == "4":  # wave M of k_backward4
    names = ["ring reads issued, T = V M (3 MFMA)", "H = C + M^T T (3 MFMA)", "gather + Quu broadcast", "-", "LDLT + solve",
             "gain stores, hand-off to G", "V_xx MFMA, next operands", "barrier"]
else:                                   # wave M of k_backward2
    names = ["operand reads (LDS) issued", "T = V M (3 MFMA)", "H = C + M^T T (3 MFMA)", "-", "gather + Quu broadcast",
             "LDLT + solve", "stores, hand-off to G, V_xx MFMA", "barrier"]
tot = med.sum()
for n_, m in zip(names, med):
    print(f"{n_:34s} {m:8.0f} cycles/knot  {100 * m / tot:5.1f} %")
print(f"{'total':34s} {tot:8.0f} cycles/knot")
