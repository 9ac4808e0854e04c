#!/usr/bin/env python3
"""Diagnostic (round 6): one launch of each k_backward4 form by how many trajectories of a block are live (B = 1..4: one block; HIP events on the
dispatch): the fused form (force_general = 5: what k_round contains), the six-wavefront form with the matrix wavefronts factoring (8) and with
the gradient wavefront factoring (7).  usage (repository root): PYTHONPATH=. python3 profiles/microbench/bw_forms_live.py"""
import os, sys
import numpy as np
from quadrotorilqr_amd import capi, problems as pb
if len(sys.argv) > 1:  # another build of the library (make variant NAME=...)
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), f'libquadrotor_ilqr_{sys.argv[1]}.so')
N = 100
for B in (1, 2, 3, 4, 64, 256):
    cfg = pb.config2(B=B, N=N)
    row = []
    for fg in (5, 8, 7):
        s = capi.from_config(cfg, force_general=fg, profile=3)
        trajs = s.forward_sim(cfg["init"], np.zeros((B, N, 52)), 1.0)
        for _ in range(3):
            s.backwards_pass(trajs)
        s.profile_reset()
        for _ in range(20):
            s.backwards_pass(trajs)
        p = s.profile_get()
        row.append(p['backward_ms'] * 1e3 / max(p['backward_launches'], 1))
        s.close()
    print(f"B={B:4d}: fused {row[0]:7.2f} us   six wavefronts, M factors {row[1]:7.2f}   six wavefronts, G factors {row[2]:7.2f}")
