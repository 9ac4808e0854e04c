#!/usr/bin/env python3
"""Diagnostic: per-knot cycle shares of the two wavefronts of k_rollout16 (separate -DQILQR_STAMPS build)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quadrotorilqr_amd import capi, problems as pb  # noqa: E402
capi.LIB_PATH = os.path.join(ROOT, "quadrotorilqr_amd", "lib", "libquadrotor_ilqr_stamps.so")
B, N = 1024, 100
cfg = pb.config2(B=B, N=N)
s = capi.from_config(cfg, single_wave_rollout=3, profile=2)
tr = cfg["init"]
for _ in range(3):
    gains, _ = s.backwards_pass(tr)
    tr_prev, tr = tr, s.forward_sim(tr, gains, 1.0)
gains, _ = s.backwards_pass(tr_prev)
for _ in range(3):
    s.forward_sim(tr_prev, gains, 1.0)
s.profile_reset()
for _ in range(5):
    s.forward_sim(tr_prev, gains, 1.0)
p = s.profile_get()
print("stamps build: k_rollout16 %.2f us per launch" % (1e3 * p["rollout_ms"] / p["rollout_launches"]))
out = np.zeros((B, 8), dtype=np.uint64)
capi.load().qilqr_debug_stamps(s._h, out.ctypes.data_as(C.c_void_p), C.c_int32(B))
blocks = B // 4
blocks = B // 4
st = out.reshape(-1)[: blocks * 24].reshape(blocks, 3, 8).astype(np.float64)
xn = ["operand reads, requests, pose stores", "Log", "wait for v_i", "control, velocity, hand-off, store", "Exp", "wait for T_{i+1}", "compose, hand-off"]
names = {0: xn, 1: xn, 2: ["requests", "wait for loads + operand registers", "wait for a free slot", "LDS writes, flag, time store"]}
for role in (0, 1, 2):
    med = np.median(st[:, role, :], axis=0) / (N / 2 if role < 2 else N)
    print("wave", ["X_0 (even knots)", "X_1 (odd knots)", "P (operands)"][role], " total %.0f cycles per %s" % (med.sum(), "own knot" if role < 2 else "knot"))
    for n_, m in zip(names[role], med):
        print(f"   {n_:42s} {m:8.0f}  {100 * m / med.sum():5.1f} %")
