"""k_backward_rollout -- the backward pass and the rollout of a round in ONE launch, up to 1024 trajectories -- against the
two kernels launched apart (QILQR_FUSE_BACKWARD_ROLLOUT=0 in the environment, read once per process: hence two child
processes).  The combined kernel contains the two kernels' bodies as statements (backward4_body.inc, rollout16_body.inc):
the same instructions on the same operands, so every output must be the same BITS -- fp64 and mixed precision, ragged
batches, per-problem desired trajectories, few-trial line searches with Levenberg-Marquardt restarts, the host-buffer path
with its copy-back under the tail.
k_round (round 4) adds the linearisation of the block's candidates to the same launch (QILQR_ROUND_KERNEL=0 keeps k_linearize a launch of
its own): se3_math.h forms its fused multiply-adds from the source alone, so the records -- and with them everything -- are the same bits
whichever kernel wrote them; a solve changes between the two forms from round to round (full blocks / one candidate per block)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from quadrotorilqr_amd import capi, problems as pb
out = {}
def keep(tag, o):
    for k in ("traj", "cost", "status", "iters", "n_bwd", "n_fwd"):
        out[tag + "_" + k] = o[k]
for B, n, seed in [(1024, 100, 2), (203, 60, 9), (5, 7, 1), (64, 30, 4)]:
    cfg = pb.config2(B=B, N=n, seed=seed)
    keep("f64_%%d" %% B, capi.from_config(cfg).solve_batch(cfg["init"]))
cfg = pb.config2(B=130, N=40, seed=5)
keep("f32", capi.from_config(cfg, precision="f32").solve_batch(cfg["init"]))
r = np.random.default_rng(3)
des = np.repeat(cfg["desired"][None], 130, axis=0)
des[:, :, 1:4] += r.uniform(-0.2, 0.2, (130, 1, 3))
keep("desired", capi.from_config(cfg).solve_batch(cfg["init"], des))
cfg = pb.config2(B=77, N=50, seed=8)
cfg["options"] = dict(cfg["options"], ls_max_iters=1)
s = capi.from_config(cfg)
s.set_regularisation(1.0, 4.0, 1e6)
keep("restarts", s.solve_batch(cfg["init"]))
np.savez(sys.argv[1], **out)
""" % ROOT


def run_child(tmp_path, fuse, round_kernel=1):
    path = os.path.join(str(tmp_path), "fuse%d%d.npz" % (fuse, round_kernel))
    env = dict(os.environ, QILQR_FUSE_BACKWARD_ROLLOUT=str(fuse), QILQR_ROUND_KERNEL=str(round_kernel))
    subprocess.run([sys.executable, "-c", CHILD, path], check=True, env=env, timeout=600)
    return np.load(path)


def test_one_launch_for_backward_and_rollout_gives_the_same_bits(tmp_path):
    apart, fused, whole = run_child(tmp_path, 0), run_child(tmp_path, 1, 0), run_child(tmp_path, 1, 1)
    assert set(apart.files) == set(fused.files) == set(whole.files) and len(apart.files) == 7 * 6
    for k in apart.files:
        np.testing.assert_array_equal(fused[k], apart[k], err_msg=k)
        np.testing.assert_array_equal(whole[k], apart[k], err_msg="k_round: " + k)
    assert np.isin(fused["f64_1024_status"], [0, 1]).all() and (fused["restarts_n_bwd"] > fused["restarts_iters"] + 1).any()
