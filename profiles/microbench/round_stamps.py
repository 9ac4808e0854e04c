#!/usr/bin/env python3
"""Diagnostic (round 6): where a k_round round's time goes, by phase, from wavefront 0's clock at the phase boundaries (-DQILQR_ROUND_STAMPS build:
hipcc ... -DQILQR_ROUND_STAMPS -o quadrotorilqr_amd/lib/libquadrotor_ilqr_roundstamps.so quadrotorilqr_amd/csrc/ilqr_capi.hip).  Per block, summed over
the rounds of a solve in which the block rolled something out: backward pass | rollout until the even step wavefront is through | behind it.
usage: PYTHONPATH=. python3 profiles/microbench/round_stamps.py [B]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quadrotorilqr_amd import capi, problems as pb  # noqa: E402
capi.LIB_PATH = os.path.join(ROOT, "quadrotorilqr_amd", "lib", "libquadrotor_ilqr_roundstamps.so")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = pb.config2(B=B, N=100, seed=2)
s = capi.from_config(cfg)
s.solve_batch(cfg["init"])
lib = capi.load()
zero = np.zeros((max(B, 8) * 8,), dtype=np.uint64)
# (the array is summed into by every solve: read it before and after one solve)
lib.qilqr_debug_stamps(s._h, zero.ctypes.data_as(C.c_void_p), C.c_int32(B))
out0 = zero.copy()
res = s.solve_batch(cfg["init"])
lib.qilqr_debug_stamps(s._h, zero.ctypes.data_as(C.c_void_p), C.c_int32(B))
d = (zero - out0).reshape(-1, 32)[: (B + 3) // 4, :4].astype(np.float64)
rounds = d[:, 3]
ok = rounds > 0
live_rounds = np.add.reduceat(np.asarray(res["n_fwd"], dtype=np.float64), np.arange(0, B, 4))   # trajectory-rounds of the block
print(f"B = {B}: blocks {ok.sum()}, rounds per block (median) {np.median(rounds[ok]):.0f}, in ticks of s_memtime / 100")
for name, k in (("backward pass", 0), ("rollout (until X_0 is through)", 1), ("behind the rollout", 2)):
    per = d[ok, k] / rounds[ok]
    print(f"  {name:32s} median {np.median(per) / 100:7.2f} us per round   (p10 {np.percentile(per, 10) / 100:.2f}, p90 {np.percentile(per, 90) / 100:.2f})")
dense = live_rounds[ok] / rounds[ok]
for lo, hi in ((0.9, 1.5), (1.5, 2.5), (2.5, 4.1)):
    m = (dense >= lo) & (dense < hi)
    if m.sum():
        tot = d[ok][m, :3].sum(axis=1) / rounds[ok][m]
        print(f"  blocks with {lo:.1f}-{hi:.1f} running trajectories per round on average: {m.sum():4d} blocks, {np.median(tot) / 100:7.2f} us per round "
              f"(backward {np.median(d[ok][m, 0] / rounds[ok][m]) / 100:.2f}, rollout {np.median(d[ok][m, 1] / rounds[ok][m]) / 100:.2f}, behind {np.median(d[ok][m, 2] / rounds[ok][m]) / 100:.2f})")
