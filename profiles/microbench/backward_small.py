#!/usr/bin/env python3
"""Diagnostic: launch time of k_backward4 and k_backward2 with very few trajectories (the late rounds of a solve):
B = 1 is one matrix wave + one gradient wave in either kernel.
usage (from the repository root): PYTHONPATH=. python3 profiles/microbench/backward_small.py"""
import numpy as np
from quadrotorilqr_amd import capi, problems as pb
for kern, name in ((4, "k_backward4"), (3, "k_backward2"), (2, "k_backward<sym>")):
    line = []
    for B in (1, 2, 4, 8, 16, 64):
        cfg = pb.config2(B=B, N=100)
        s = capi.from_config(cfg, profile=2, force_general=kern)
        traj = s.forward_sim(cfg["init"], np.zeros((B, 100, 52)), 1.0)
        for _ in range(3): s.backwards_pass(traj)
        s.profile_reset()
        for _ in range(8): s.backwards_pass(traj)
        p = s.profile_get()
        line.append(f"B={B}: {1e3 * p['backward_ms'] / p['backward_launches']:.1f}")
        s.close()
    print(name, "us/launch  ", "  ".join(line), flush=True)
