"""Generates tests/golden/oracle_golden.npz from the CPU oracle (SELF-goldens).

Which goldens are what: oracle_golden.npz (this script) holds SELF-goldens -- the reference's C++ solver cannot be
built in this image (Eigen / manif absent), so these vectors come from oracle/ilqr_oracle.c; reference_demo_inputs.npz
(make_reference_fixture.py, beside this file) holds REFERENCE-DERIVED vectors -- computed by the reference's own Python,
quadrotor_ilqr.py, imported in the build container (tests/test_reference_fixture.py).

The vectors here come from oracle/ilqr_oracle.c after it passed the reference's own
known-answer and finite-difference tests (tests/test_oracle_*.py) and the scipy
expm/logm cross-checks.  They pin the oracle against regressions and give the GPU
tests fixed inputs/outputs.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as orc  # noqa: E402
from quadrotorilqr_amd import problems as pb  # noqa: E402

out = {}

# (i) per-function vectors, including small-angle, near-pi and w<0 cases
rng = np.random.default_rng(11)
taus = np.array([
    [1, 2, 3, 0.4, 0.5, 0.6], [0.3, 0.6, 0.9, 1.2, 1.5, 1.8], [0.3, -0.2, 0.1, 1e-3, -2e-3, 1.5e-3],
    [0.3, -0.2, 0.1, 1e-6, -2e-6, 1.5e-6], [0.3, -0.2, 0.1, 0, 0, 0], [-1, 0.5, 2, 3.1, 0, 0],
    [-1, 0.5, 2, 0, 0, np.pi - 1e-4]] + [np.concatenate([rng.uniform(-2, 2, 3), rng.uniform(-1.7, 1.7, 3)]) for _ in range(9)])
out["lie_tau"] = taus
out["lie_exp"] = np.array([orc.se3_exp(t) for t in taus])
out["lie_log_of_exp"] = np.array([orc.se3_log(orc.se3_exp(t)) for t in taus])
out["lie_rjac"] = np.array([orc.se3_rjac(t) for t in taus])
out["lie_rjacinv"] = np.array([orc.se3_rjacinv(t) for t in taus])
out["lie_adj_exp"] = np.array([orc.se3_adj(orc.se3_exp(t)) for t in taus])

# (ii) per-knot differentials at the reference's P-test point (quadrotor_model_test.cc:152-157)
A = np.random.default_rng(0).uniform(-1, 1, (3, 3))
inertia = A @ A.T + 3 * np.eye(3)
mp = orc.model_params(1.0, inertia, 1.0, 1.0, 9.81)
x = np.concatenate([orc.se3_exp([1, 2, 3, 4, 5, 6.0]), [2, 3, 4, 5, 6, 7.0]])
xd = np.concatenate([orc.se3_exp(2 * np.arange(1, 7.0)), 2 * np.arange(2, 8.0)])
u = np.array([1.0, 2, 3, 4])
xn, Jx, Ju = orc.discrete_dynamics(mp, x, u, 0.1, diffs=True)
Qd = rng.uniform(-1, 1, (12, 12)) + 6 * np.eye(12)
Rd = rng.uniform(-1, 1, (4, 4)) + 3 * np.eye(4)
c, D = orc.cost(Qd, Rd, x, u, xd, np.zeros(4), diffs=True)
out.update(knot_inertia=inertia, knot_x=x, knot_xd=xd, knot_u=u, knot_xnext=xn, knot_Jx=Jx, knot_Ju=Ju,
           knot_Q=Qd, knot_R=Rd, knot_cost=c, knot_Cx=D["x"], knot_Cu=D["u"], knot_Cxx=D["xx"])

# (iii) full solve traces
for name, H in (("demo40", 4.0), ("demo100", 10.0)):
    d = pb.box_climb_desired(H)
    s = orc.OracleSolver(orc.model_params(**pb.MODEL_D), pb.Q_DEMO, pb.R_DEMO, d, pb.DT_DEMO,
                         orc.options(**pb.OPTIONS_DEMO))
    o = s.solve(d)
    out[name + "_desired"] = d
    out[name + "_traj"] = o["traj"]
    out[name + "_cost_hist"] = o["cost_hist"]
    out[name + "_meta"] = np.array([o["status"], o["iters"], o["n_bwd"], o["n_fwd"]])
    g, terms = s.backwards_pass(d)
    out[name + "_gains0"] = g
    out[name + "_terms0"] = terms

cfg = pb.config2(B=8)
s = orc.OracleSolver(orc.model_params(**cfg["model"]), cfg["Q"], cfg["R"], cfg["desired"], cfg["dt"],
                     orc.options(**cfg["options"]))
o = s.solve_batch(cfg["init"])
out.update(cfg2_init=cfg["init"], cfg2_desired=cfg["desired"], cfg2_traj=o["traj"], cfg2_cost=o["cost"],
           cfg2_status=o["status"], cfg2_iters=o["iters"], cfg2_n_bwd=o["n_bwd"], cfg2_n_fwd=o["n_fwd"])
hist = [s.solve(cfg["init"][b])["cost_hist"] for b in range(8)]
out["cfg2_cost_hist"] = np.array([np.pad(h, (0, 101 - len(h)), constant_values=np.nan) for h in hist])

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
