"""Diagnostic: the general kernel's final costs at 200 knots (the unsymmetrised recursion's rounding garbage) over the symmetric kernels' costs."""
import numpy as np
from quadrotorilqr_amd import capi, problems as pb
cfg = pb.config3(B=16, N=200)
sym = capi.from_config(cfg).solve_batch(cfg["init"])
gen = capi.from_config(cfg, force_general=True).solve_batch(cfg["init"])
np.set_printoptions(linewidth=200, precision=3)
print("sym", sym["cost"]); print("gen", gen["cost"]); print("ratio", gen["cost"] / sym["cost"]); print("status", gen["status"], gen["iters"])
