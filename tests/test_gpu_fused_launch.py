"""k_backward_rollout -- the backward pass and the rollout of a round in ONE launch, up to 1024 trajectories -- against the
two kernels launched apart (qilqr_device_config.round_launch = 1; until ABI version 7 an environment variable read once per
process, hence child processes then).  The combined kernel contains the two kernels' bodies as statements
(backward4_body.inc, rollout16_body.inc): the same instructions on the same operands, so every output must be the same BITS
-- fp64 and mixed precision, ragged batches, per-problem desired trajectories, few-trial line searches with
Levenberg-Marquardt restarts, the host-buffer path with its copy-back under the tail.
k_round (round 4) adds the linearisation of the block's candidates to the same launch (round_launch = 2 keeps k_linearize a launch
of its own): se3_math.h forms its fused multiply-adds from the source alone, so the records -- and with them everything -- are the
same bits whichever kernel wrote them; a solve changes between the two forms from round to round (full blocks / one candidate per
block).  rounds_per_launch (1, 2, 4 rounds in one k_round launch) changes no bit either."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from quadrotorilqr_amd import capi, problems as pb  # noqa: E402


def solves(**kw):
    out = {}

    def keep(tag, o):
        for k in ("traj", "cost", "status", "iters", "n_bwd", "n_fwd"):
            out[tag + "_" + k] = o[k]

    for B, n, seed in [(1024, 100, 2), (203, 60, 9), (5, 7, 1), (64, 30, 4)]:
        cfg = pb.config2(B=B, N=n, seed=seed)
        keep("f64_%d" % B, capi.from_config(cfg, **kw).solve_batch(cfg["init"]))
    cfg = pb.config2(B=130, N=40, seed=5)
    keep("f32", capi.from_config(cfg, precision="f32", **kw).solve_batch(cfg["init"]))
    r = np.random.default_rng(3)
    des = np.repeat(cfg["desired"][None], 130, axis=0)
    des[:, :, 1:4] += r.uniform(-0.2, 0.2, (130, 1, 3))
    keep("desired", capi.from_config(cfg, **kw).solve_batch(cfg["init"], des))
    cfg = pb.config2(B=77, N=50, seed=8)
    cfg["options"] = dict(cfg["options"], ls_max_iters=1)
    s = capi.from_config(cfg, **kw)
    s.set_regularisation(1.0, 4.0, 1e6)
    keep("restarts", s.solve_batch(cfg["init"]))
    return out


def test_one_launch_for_backward_and_rollout_gives_the_same_bits():
    apart, fused, whole = solves(round_launch=1), solves(round_launch=2), solves()
    assert set(apart) == set(fused) == set(whole) and len(apart) == 7 * 6
    for k in apart:
        np.testing.assert_array_equal(fused[k], apart[k], err_msg=k)
        np.testing.assert_array_equal(whole[k], apart[k], err_msg="k_round: " + k)
    assert np.isin(fused["f64_1024_status"], [0, 1]).all() and (fused["restarts_n_bwd"] > fused["restarts_iters"] + 1).any()


@pytest.mark.parametrize("n", [1, 2, 3, 15, 16, 17, 18, 31, 32, 33, 64, 65, 127])
def test_linearisation_behind_the_rollout_at_horizons_around_its_chunk_size(n):
    """Round 6: k_round's other wavefronts linearise the candidates' knots BEHIND the rollout, sixteen knots x four rows to a task, as the step
    wavefronts announce them (round_follow; one running trajectory per block: sixty-four knots to a task, after the rollout).  Horizons either
    side of the chunk boundaries, ragged batches whose last block holds one, two or three trajectories, blocks in which trajectories finish in
    different rounds: the same bits as three launches per round, every array."""
    for B in (1, 6, 7, 9):
        cfg = pb.config2(B=B, N=n, seed=30 + B)
        a = capi.from_config(cfg).solve_batch(cfg["init"])
        b = capi.from_config(cfg, round_launch=1).solve_batch(cfg["init"])
        six = capi.from_config(cfg, force_general=8).solve_batch(cfg["init"])
        for k in ("traj", "cost", "status", "iters", "n_bwd", "n_fwd"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=f"B={B} n={n} {k}")
            np.testing.assert_array_equal(six[k], b[k], err_msg=f"six wavefronts, B={B} n={n} {k}")
        assert (a["status"] >= 0).all() and np.isfinite(a["traj"]).all()


@pytest.mark.parametrize("rounds", [0, 1])
def test_k_round_with_the_six_wavefront_backward_pass_gives_the_same_bits(rounds):
    """Round 6: k_round<.., SIX> -- the same launch with the backward pass in the six-wavefront form (the matrix wavefronts factor, knot loop
    unrolled), which the host takes once at most two trajectories per block run on average (5.7 us of a 127 us round with one running
    trajectory per block).  force_general = 8 takes it in EVERY launch, 5 never; the automatic choice changes over inside a solve.  Four
    rounds per launch and one; every case of `solves` (ragged batches, per-problem desired trajectories, restarts; the mixed mode keeps
    k_backward_rollout) and the profile says that the launches were k_round's (no separate rollout launches)."""
    six, fused, auto = solves(force_general=8, rounds_per_launch=rounds), solves(force_general=5, rounds_per_launch=rounds), solves(rounds_per_launch=rounds)
    for k in fused:
        np.testing.assert_array_equal(six[k], fused[k], err_msg="six wavefronts: " + k)
        np.testing.assert_array_equal(auto[k], fused[k], err_msg="automatic: " + k)
    cfg = pb.config2(B=1024, N=100, seed=2)
    s = capi.from_config(cfg, force_general=8, profile=2, rounds_per_launch=rounds)
    s.solve_batch(cfg["init"])
    p = s.profile_get()
    assert p["rollout_launches"] == 0 and p["backward_launches"] > 0 and "k_round" in s.describe(1024)


@pytest.mark.parametrize("rounds", [1, 2])
def test_rounds_per_launch_changes_no_bit(rounds):
    """(qilqr_solve_batch from pageable arrays: staging, then the device-resident solve whose launches may hold several rounds)"""
    cfg = pb.config2(B=1024, N=100, seed=2)
    a = capi.from_config(cfg).solve_batch(cfg["init"])
    b = capi.from_config(cfg, rounds_per_launch=rounds).solve_batch(cfg["init"])
    for k in ("traj", "cost", "status", "iters", "n_bwd", "n_fwd"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_bad_round_launch_values_are_refused():
    cfg = pb.config2(B=4, N=8)
    with pytest.raises(TypeError, match="round_launch"):
        capi.from_config(cfg, round_launch=3)
    with pytest.raises(TypeError, match="rounds_per_launch"):
        capi.from_config(cfg, rounds_per_launch=3)
