#!/usr/bin/env python3
"""Diagnostic: average k_linearize launch time through qilqr_cost_trajectory (profile = 2).
usage: [BLIST=16,1024] linearize_time.py [lib.so ...]   (B = 16: the lone-wave time of the longer half plus launch)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quadrotorilqr_amd import capi, problems as pb  # noqa: E402

libs = sys.argv[1:] or [capi.LIB_PATH]
for lib in libs:
    capi.LIB_PATH = os.path.abspath(lib)
    capi._lib = None
    for B in [int(x) for x in os.environ.get("BLIST", "1024,8192").split(",")]:
        cfg = pb.config2(B=B, N=100)
        s = capi.from_config(cfg, profile=2)
        traj = s.forward_sim(cfg["init"], np.zeros((B, 100, 52)), 1.0)
        for _ in range(3):
            s.cost_trajectory(traj)
        s.profile_reset()
        for _ in range(10):
            s.cost_trajectory(traj)
        p = s.profile_get()
        print(os.path.basename(lib), "B", B, "k_linearize us/launch", round(1e3 * p["linearize_ms"] / p["linearize_launches"], 2))
        s.close()
