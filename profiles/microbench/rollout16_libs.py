#!/usr/bin/env python3
"""Diagnostic: launch time of k_rollout16 (a solve's third iteration, all 1024 trajectories live) for several builds.
usage (from the repository root): python profiles/microbench/rollout16_libs.py lib1.so [lib2.so ...]"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from quadrotorilqr_amd import capi, problems as pb
B = 1024
cfg = pb.config2(B=B, N=100)
base = capi.from_config(cfg)
tr = cfg["init"]
for _ in range(3):
    gains, _ = base.backwards_pass(tr)
    tr_prev, tr = tr, base.forward_sim(tr, gains, 1.0)
gains, _ = base.backwards_pass(tr_prev)
base.close()
for lib in sys.argv[1:]:
    capi.LIB_PATH = os.path.abspath(lib); capi._lib = None
    s = capi.from_config(dict(cfg, init=tr_prev), profile=2, single_wave_rollout=3)
    for _ in range(3): s.forward_sim(tr_prev, gains, 1.0)
    s.profile_reset()
    for _ in range(10): s.forward_sim(tr_prev, gains, 1.0)
    p = s.profile_get()
    print(os.path.basename(lib), "k_rollout16 us/launch", round(1e3 * p["rollout_ms"] / p["rollout_launches"], 2), flush=True)
    s.close()
