#!/usr/bin/env python3
"""Diagnostic: where the eight wavefronts of k_solve4 spend their cycles (separate -DQILQR_STAMPS build)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quadrotorilqr_amd import capi, problems as pb  # noqa: E402
capi.LIB_PATH = os.path.join(ROOT, "quadrotorilqr_amd", "lib", "libquadrotor_ilqr_stamps.so")
B, N = 1024, 100
cfg = pb.config2(B=B, N=N)
s = capi.from_config(cfg, persistent=1)
for _ in range(3):
    out = s.solve_batch(cfg["init"])
st = np.zeros((B, 8), dtype=np.uint64)
capi.load().qilqr_debug_stamps(s._h, st.ctypes.data_as(C.c_void_p), C.c_int32(B))
W = 8  # wavefronts per block
blocks = (B * 8) // (W * 8)  # the stamp buffer holds B x 8 words: the first 128 blocks of 256
st = st.reshape(-1)[: blocks * W * 8].reshape(blocks, W, 8).astype(np.float64)
names = ["first linearisation", "settle", "backward", "forward: own role", "forward: at the closing barrier"]
it = st[:, 0, 5]
print("iterations per block: mean %.1f max %d" % (it.mean(), it.max()))
worst = int(np.argmax(st[:, 0, :5].sum(axis=1)))
for w in range(W):
    tot = st[:, w, :5].sum(axis=1)
    print("wave %d: block total cycles median %.0f max %.0f   (slowest block, per iteration:)" % (w, np.median(tot), tot.max()),
          "  ".join("%s %.0f" % (n_, st[worst, w, k] / max(st[worst, w, 5], 1)) for k, n_ in enumerate(names)))
