#!/usr/bin/env python3
"""Diagnostic: per-wavefront start/end of k_linearize (separate -DQILQR_STAMPS build)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quadrotorilqr_amd import capi, problems as pb  # noqa: E402

capi.LIB_PATH = os.environ.get("QLIB", os.path.join(ROOT, "quadrotorilqr_amd", "lib", "libquadrotor_ilqr_stamps.so"))
B, N = int(os.environ.get("B", "1024")), 100
cfg = pb.config2(B=B, N=N)
s = capi.from_config(cfg)
mode = sys.argv[1] if len(sys.argv) > 1 else "converged"
if mode == "first":
    trajs = s.forward_sim(cfg["init"], np.zeros((B, N, 52)), 1.0)
else:
    trajs = s.solve_batch(cfg["init"])["traj"]
for _ in range(3):
    s.cost_trajectory(trajs)
out = np.zeros((B, 8), dtype=np.uint64)
capi.load().qilqr_debug_stamps(s._h, out.ctypes.data_as(C.c_void_p), C.c_int32(B))
w = out.reshape(-1, 4).astype(np.float64)
w = w[w[:, 1] > 0]
t0 = w[:, 0].min()
print("mode", mode, "B", B, "waves recorded", len(w), "(of", 2 * ((B + 63) // 64) * N, ")")
for half in (0, 1):
    h = w[w[:, 3] == half]
    if len(h) == 0:
        continue
    print(["dynamics half", "cost half"][half], ": start us (min/med/max) %.2f %.2f %.2f   duration us (min/med/max) %.2f %.2f %.2f   cycles med %.0f"
          % ((h[:, 0].min() - t0) / 100, (np.median(h[:, 0]) - t0) / 100, (h[:, 0].max() - t0) / 100,
             (h[:, 1] - h[:, 0]).min() / 100, np.median(h[:, 1] - h[:, 0]) / 100, (h[:, 1] - h[:, 0]).max() / 100,
             np.median(h[:, 2])))
print("last end us %.2f" % ((w[:, 1].max() - t0) / 100))
