#!/usr/bin/env python3
"""Diagnostic: average launch time of the backward kernel through qilqr_backwards_pass (profile = 2, all 1024
trajectories running), for one or more builds of the library.  usage: backward_time.py lib1.so [lib2.so ...]
(run from the repository root)"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from quadrotorilqr_amd import capi, problems as pb
for lib in sys.argv[1:]:
    capi.LIB_PATH = os.path.abspath(lib); capi._lib = None
    cfg = pb.config2(B=1024, N=100)
    s = capi.from_config(cfg, profile=2)
    traj = s.forward_sim(cfg["init"], np.zeros((1024, 100, 52)), 1.0)
    for _ in range(3): s.backwards_pass(traj)
    s.profile_reset()
    for _ in range(10): s.backwards_pass(traj)
    p = s.profile_get()
    print(os.path.basename(lib), "k_backward us/launch", round(1e3 * p["backward_ms"] / p["backward_launches"], 2))
