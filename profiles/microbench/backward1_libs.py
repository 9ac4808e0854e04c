#!/usr/bin/env python3
"""Diagnostic: launch time of the one-wavefront backward kernel (force_general = 2) with every trajectory live, for
several builds.  usage (from the repository root): python profiles/microbench/backward1_libs.py B lib1.so [lib2.so ...]"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from quadrotorilqr_amd import capi, problems as pb
B = int(sys.argv[1])
for lib in sys.argv[2:]:
    capi.LIB_PATH = os.path.abspath(lib); capi._lib = None
    cfg = pb.config2(B=B, N=100)
    s = capi.from_config(cfg, profile=2, force_general=2)
    traj = s.forward_sim(cfg["init"], np.zeros((B, 100, 52)), 1.0)
    for _ in range(2): s.backwards_pass(traj)
    s.profile_reset()
    for _ in range(5): s.backwards_pass(traj)
    p = s.profile_get()
    print(os.path.basename(lib), "B", B, "k_backward us/launch", round(1e3 * p["backward_ms"] / p["backward_launches"], 2), flush=True)
    s.close()
