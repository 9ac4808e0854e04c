#!/usr/bin/env python3
"""Diagnostic: average launch time of the rollout kernel through qilqr_forward_sim (profile = 2, 1024
trajectories), for one or more builds of the library.  usage: rollout_time.py lib1.so [lib2.so ...]
(run from a directory three levels below the repository root)"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd() + "/../../..")
from quadrotorilqr_amd import capi, problems as pb
for lib in sys.argv[1:]:
    capi.LIB_PATH = os.path.abspath(lib); capi._lib = None
    cfg = pb.config2(B=1024, N=100)
    s = capi.from_config(cfg, profile=2)
    out = s.solve_batch(cfg["init"])
    gains, _ = s.backwards_pass(out["traj"])
    for _ in range(3): s.forward_sim(out["traj"], gains, 1.0)
    s.profile_reset()
    for _ in range(10): s.forward_sim(out["traj"], gains, 1.0)
    p = s.profile_get()
    print(os.path.basename(lib), "k_rollout us/launch", round(1e3 * p["rollout_ms"] / p["rollout_launches"], 2))
