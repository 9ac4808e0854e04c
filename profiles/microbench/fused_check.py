#!/usr/bin/env python3
"""Diagnostic: k_backward4 with the gradient fused into the matrix wavefronts (force_general = 5) against the six-wavefront form
(4) and the oracle: one pass and whole solves."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from quadrotorilqr_amd import capi, problems as pb
from oracle import oracle as orc
for B, n, seed in ((37, 60, 3), (5, 1, 3), (7, 2, 3), (9, 3, 3), (203, 100, 2)):
    cfg = pb.config2(B=B, N=n, seed=seed)
    a, b = capi.from_config(cfg, force_general=4), capi.from_config(cfg, force_general=5)
    trajs = a.forward_sim(cfg["init"], np.zeros((B, n, 52)), 1.0)
    ga, ta = a.backwards_pass(trajs)
    gb, tb = b.backwards_pass(trajs)
    print(B, n, "pass: max |dgain| / max|gain| %.2e, terms rel %.2e" % (np.abs(ga - gb).max() / np.abs(ga).max(), np.abs(ta - tb).max() / np.abs(ta).max()))
    oa, ob = a.solve_batch(cfg["init"]), b.solve_batch(cfg["init"])
    ref = orc.OracleSolver(orc.model_params(**cfg["model"]), cfg["Q"], cfg["R"], cfg["desired"], cfg["dt"], orc.options(**cfg["options"])).solve_batch(cfg["init"][:16], n_threads=8)
    print("   solve: status eq", (oa["status"] == ob["status"]).all(), "iters eq", (oa["iters"] == ob["iters"]).all(), "n_fwd eq", (oa["n_fwd"] == ob["n_fwd"]).all(),
          "cost rel %.2e" % np.max(np.abs(oa["cost"] - ob["cost"]) / np.abs(oa["cost"])), "| vs oracle: iters eq", (ob["iters"][:16] == ref["iters"]).all(),
          "cost rel %.2e traj abs %.2e" % (np.max(np.abs(ob["cost"][:16] - ref["cost"]) / np.abs(ref["cost"])), np.abs(ob["traj"][:16] - ref["traj"]).max()))
