"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs and against the committed golden fixtures.

Stated fp64 tolerances (justified by the 1e-13-perturbation experiment of
BASELINE.md section 2): one pass from identical inputs 1e-10 relative; full solve: cost history
1e-8 relative per iteration, final cost 1e-9 relative, final trajectory 1e-6 absolute (the
reference's own bar, ilqr_test.cc:189); iteration counts equal.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402  (the checker)
from quadrotorilqr_amd import capi, problems as pb
from tests.diag_lib import capi_diag  # the diagnostics build: k_solve4, k_backward2  # noqa: E402
from tests.observed import observed  # noqa: E402

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_golden.npz"))


from tests.exit_paths import assert_same_exit_paths, explain, describe  # noqa: E402  (counts equal, or the oracle's deciding margin shown)


def oracle_for(cfg, recursion=0, **opt_over):
    """recursion = 1: the oracle's substituted, symmetrised value update (orc_set_recursion; tests/test_oracle_recursion.py)"""
    s = orc.OracleSolver(orc.model_params(**cfg["model"]), cfg["Q"], cfg["R"], cfg["desired"], cfg["dt"],
                         orc.options(**dict(cfg["options"], **opt_over)))
    if recursion:
        s.set_recursion(recursion)
    return s


def random_cfg(seed, n=25, dense=False, B=6):
    r = np.random.default_rng(seed)
    A = r.uniform(-1, 1, (3, 3))
    model = dict(mass_kg=1.3, inertia=A @ A.T + 3 * np.eye(3), arm_length_m=0.7,
                 torque_to_thrust_ratio_m=0.2, g_mpss=9.81)
    if dense == "qsym":   # symmetric Q (packed record layout) with a NON-symmetric R (general backward kernel)
        Q = r.uniform(-1, 1, (12, 12))
        Q = Q @ Q.T + 12 * np.eye(12)
        R = r.uniform(-0.3, 0.3, (4, 4)) + 2 * np.eye(4)
    elif dense == "sym":  # dense symmetric Q and R (two-wavefront backward kernel, 128-entry records)
        Q = r.uniform(-1, 1, (12, 12))
        Q = Q @ Q.T + 12 * np.eye(12)
        R = r.uniform(-0.3, 0.3, (4, 4))
        R = R + R.T + 2 * np.eye(4)
    elif dense:
        Q = r.uniform(-1, 1, (12, 12))
        Q = Q @ Q.T + 12 * np.eye(12) + 0.3 * r.uniform(-1, 1, (12, 12))  # NOT symmetric
        R = r.uniform(-0.3, 0.3, (4, 4)) + 2 * np.eye(4)
    else:
        Q, R = pb.Q_DEMO, pb.R_DEMO

    def rand_traj():
        t = np.zeros((n, 18))
        t[:, 0] = 0.1 * np.arange(n)
        for i in range(n):
            t[i, 1:8] = orc.se3_exp(np.concatenate([r.uniform(-2, 2, 3), r.uniform(-1.5, 1.5, 3)]))
        t[:, 8:14] = r.uniform(-2, 2, (n, 6))
        t[:, 14:18] = r.uniform(0, 5, (n, 4))
        return t

    desired = rand_traj()
    trajs = np.stack([rand_traj() for _ in range(B)])
    return dict(model=model, Q=Q, R=R, dt=0.1, desired=desired, init=trajs,
                options=dict(pb.OPTIONS_DEMO, populate_debug=False))


# ------------------------------------------------------------------ ilqr_test.cc fixture on the GPU
N, DT = 3, 0.1


@pytest.fixture(scope="module")
def fx():
    cur = pb.identity_trajectory(N, DT)
    model = dict(mass_kg=1.0, inertia=np.eye(3), arm_length_m=1.0, torque_to_thrust_ratio_m=1.0, g_mpss=0.0)
    opts = dict(step_update=0.5, desired_reduction_frac=0.5, ls_max_iters=10, rtol=1e-12, atol=1e-12,
                max_iters=100, populate_debug=True)
    s = capi.QuadrotorILQRBatch(1.0, np.eye(3), 1.0, 1.0, 0.0, np.eye(12), np.eye(4), cur, DT, opts)
    gains = orc.kK_to_gains(np.ones((N, 4)), np.zeros((N, 4, 12)))
    return dict(s=s, cur=cur, gains=gains, model=model, opts=opts)


def traj_close(a, b, tol=1e-6):
    for pa, pb_ in zip(a, b):
        rel = orc.se3_compose(orc.se3_inverse(pa[1:8]), pb_[1:8])
        assert np.linalg.norm(orc.se3_log(rel)) < tol
        np.testing.assert_allclose(pa[8:18], pb_[8:18], atol=tol, rtol=1e-12)


def test_forward_sim_known_answer(fx):  # ilqr_test.cc:102-126
    new = fx["s"].forward_sim(fx["cur"][None], fx["gains"][None])[0]
    exp = pb.identity_trajectory(N, DT)
    exp[:, 14:18] = 1.0
    exp[1, 10] = DT * 4.0
    exp[2, 3] = DT * DT * 4.0
    exp[2, 10] = 2 * DT * 4.0
    traj_close(new, exp)
    np.testing.assert_array_equal(new[:, 0], fx["cur"][:, 0])


def test_cost_trajectory_known_answer(fx):  # ilqr_test.cc:128-141, EXPECT_DOUBLE_EQ = 4 ULP
    new = fx["s"].forward_sim(fx["cur"][None], fx["gains"][None])
    cost = fx["s"].cost_trajectory(new)[0]
    expected = (DT * 4.0) ** 2.0 + (DT * DT * 4.0) ** 2.0 + (2.0 * DT * 4.0) ** 2.0 + 3 * 4
    assert abs(cost - expected) <= 4 * np.spacing(expected)


def test_backward_pass_zero_gradient(fx):  # ilqr_test.cc:143-153: exact zeros
    gains, terms = fx["s"].backwards_pass(fx["cur"][None])
    assert terms[0, 0] == 0.0 and terms[0, 1] == 0.0
    k, _ = orc.gains_to_kK(gains[0])
    assert np.all(k == 0.0)


def test_backward_pass_negative_expected_reduction(fx):  # ilqr_test.cc:155-164
    new = fx["s"].forward_sim(fx["cur"][None], fx["gains"][None])
    _, terms = fx["s"].backwards_pass(new)
    assert terms[0, 0] < 0.0
    assert abs(terms[0, 0] - (-25.6032)) < 1e-9


def test_line_search_reduces_cost(fx):  # ilqr_test.cc:166-177
    s = fx["s"]
    traj = s.forward_sim(fx["cur"][None], fx["gains"][None])
    cost = s.cost_trajectory(traj)
    gains, terms = s.backwards_pass(traj)
    ls = s.line_search(traj, cost, gains, terms)
    assert ls["status"][0] == 0
    dj = ls["step"][0] * terms[0, 0] + ls["step"][0] ** 2 * terms[0, 1] / 2.0
    assert ls["cost"][0] - cost[0] < 0.5 * dj
    ref = oracle_for(dict(model=fx["model"], Q=np.eye(12), R=np.eye(4), desired=fx["cur"], dt=DT,
                          options=fx["opts"])).line_search(traj[0], cost[0], gains[0], terms[0])
    assert ls["step"][0] == ref["step"]
    np.testing.assert_allclose(ls["cost"][0], ref["cost"], rtol=1e-10)


def test_solve_finds_optimal_trajectory(fx):  # ilqr_test.cc:179-190
    k = np.ones((N, 4))
    k[:, 0] *= 100
    k[:, 2] *= 100
    init = fx["s"].forward_sim(fx["cur"][None], orc.kK_to_gains(k, np.zeros((N, 4, 12)))[None])[0]
    traj, info = fx["s"].solve(init)
    traj_close(fx["cur"], traj, 1e-6)
    assert info["cost"] < 1e-20
    assert len(info["debug_costs"]) == info["iters"]
    np.testing.assert_array_equal(info["debug_trajs"][-1], traj)


# ------------------------------------------------------------------ per-pass parity on random inputs
@pytest.mark.parametrize("seed,dense", [(11, False), (12, True), (13, True), (14, "qsym"), (15, "sym")])
def test_passes_match_oracle(seed, dense):
    cfg = random_cfg(seed, dense=dense)
    s = capi.from_config(cfg)
    ref = oracle_for(cfg)
    trajs = cfg["init"]
    cost = s.cost_trajectory(trajs)
    gains, terms = s.backwards_pass(trajs)
    for b in range(len(trajs)):
        np.testing.assert_allclose(cost[b], ref.cost_trajectory(trajs[b]), rtol=1e-12)
        g_ref, t_ref = ref.backwards_pass(trajs[b])
        np.testing.assert_allclose(terms[b], t_ref, rtol=1e-10)
        np.testing.assert_allclose(gains[b], g_ref, rtol=1e-9, atol=1e-10 * np.abs(g_ref).max())
    r = np.random.default_rng(seed)
    small_gains = 0.05 * r.uniform(-1, 1, gains.shape)
    alpha = np.array([1.0, 0.5, 0.25, 1.0, 0.125, 1.0])
    out = s.forward_sim(trajs, small_gains, alpha)
    for b in range(len(trajs)):
        np.testing.assert_allclose(out[b], ref.forward_sim(trajs[b], small_gains[b], alpha[b]), rtol=1e-10,
                                   atol=1e-10)


@pytest.mark.parametrize("n", [100, 150])
def test_general_kernel_pass_at_long_horizons_stays_within_the_recursion_s_own_sensitivity(n):
    """the general kernel (force_general = 1: the reference's forms, V and V^T alternating in the accumulator) against the oracle's
    reference form, ONE backward pass at 100 and 150 knots, knot by knot.  The unsymmetrised recursion amplifies rounding by ~1.3 per knot
    from the horizon's end, so the comparison is (a) tight where the amplification is still small -- the last 60 knots:
    1e-9 of the largest gain --, and (b) everywhere within 100 x the envelope of what the SAME oracle source shows between its own two
    arithmetics (with and without fused multiply-adds): the kernel is as close to the reference form as the reference form is to itself.
    Symmetric weights (the hover problem of configs[1]) and non-symmetric Q at the asymmetry the recursion still tolerates (1e-13)."""
    U = np.triu(np.random.default_rng(7).uniform(-1, 1, (12, 12)), 1)
    for eps in (0.0, 1e-13):
        cfg = pb.config2(B=6, N=n, seed=5)
        cfg["Q"] = cfg["Q"] + eps * U
        s = capi.from_config(cfg, force_general=1)
        assert "general" in s.describe(6)
        P = oracle_for(cfg)
        F = orc.OracleSolver(orc.model_params(**cfg["model"]), cfg["Q"], cfg["R"], cfg["desired"], cfg["dt"], orc.options(**cfg["options"]),
                             library=orc.fast_library(native=False))
        trajs = s.forward_sim(cfg["init"], np.zeros((6, n, 52)), 1.0)
        g, tm = s.backwards_pass(trajs)
        for b in range(6):
            gp, tp = P.backwards_pass(trajs[b])
            gf, _ = F.backwards_pass(trajs[b])
            scale = np.abs(gp).max()
            np.testing.assert_allclose(g[b][n - 60:], gp[n - 60:], rtol=0, atol=1e-9 * scale)
            own = np.abs(gf - gp).max(axis=1)
            env = np.maximum.accumulate(own[::-1])[::-1]           # what rounding has grown to by knot k, counted from the end
            diff = np.abs(g[b] - gp).max(axis=1)
            bad = diff > 100.0 * env + 1e-10 * scale
            assert not bad.any(), (eps, b, int(np.argmax(bad)), diff[bad][:3], env[bad][:3])
            np.testing.assert_allclose(tm[b], tp, rtol=1e-6 if n == 100 else 1e-2)


def test_zero_cost_and_zero_gradient_at_zero_error():
    """cost_test.cc:27-39 (EXPECT_EQ(cost, 0.0) at a random pose) and ilqr_test.cc:143-153 on the
    device: x (-) x is exactly zero for arbitrary unit quaternions, so cost, C_x, k and both
    reduction terms are exact zeros."""
    cfg = random_cfg(31, n=12, B=3)
    d = cfg["desired"].copy()
    d[:, 14:18] = 0.0
    cfg = dict(cfg, desired=d, model=dict(cfg["model"], g_mpss=0.0))  # zero control is an equilibrium input only at g = 0
    d2 = d.copy()
    d2[:, 8:14] = 0.0   # at rest: the desired trajectory is then a fixed point of the dynamics only if constant
    s = capi.from_config(dict(cfg, desired=d))
    assert (s.cost_trajectory(np.stack([d, d, d])) == 0.0).all()
    gains, terms = s.backwards_pass(d[None])
    assert terms[0, 0] == 0.0 and terms[0, 1] == 0.0
    k, _ = orc.gains_to_kK(gains[0])
    assert np.all(k == 0.0)
    # the demo's desired trajectory (roll pi/3, 2pi/3, pi quaternions): same property
    cfg1 = pb.config1(4.0)
    s1 = capi.from_config(cfg1)
    assert s1.cost_trajectory(cfg1["init"])[0] == 0.0
    _, t1 = s1.backwards_pass(cfg1["init"])
    assert t1[0, 0] == 0.0 and t1[0, 1] == 0.0


def test_symmetric_fast_path_agrees_with_general_path():
    """k_backward<SYM> (no transpose, symmetric weights) against k_backward<general> on the same data"""
    cfg = pb.config2(B=32, N=60)
    fast = capi.from_config(cfg)
    gen = capi.from_config(cfg, force_general=True)
    trajs = fast.forward_sim(cfg["init"], np.zeros((32, 60, 52)), 1.0)  # a feasible rollout
    gf, tf = fast.backwards_pass(trajs)
    gg, tg = gen.backwards_pass(trajs)
    np.testing.assert_allclose(tf, tg, rtol=1e-11)
    np.testing.assert_allclose(gf, gg, rtol=1e-7, atol=1e-9 * np.abs(gg).max())
    of, og = fast.solve_batch(cfg["init"]), gen.solve_batch(cfg["init"])
    np.testing.assert_array_equal(of["iters"], og["iters"])
    np.testing.assert_allclose(of["cost"], og["cost"], rtol=1e-10)
    np.testing.assert_allclose(of["traj"], og["traj"], atol=1e-7)


def test_two_wave_backward_matches_single_wave():
    """k_backward2 (matrix wave + gradient wave, operands streamed through LDS) performs the operations of
    k_backward<SYM> with the gradient recursion moved to its own wavefront: same gains and terms to rounding
    (the compiler may contract multiply-adds differently), same solves.  Horizons 1, 2, 3 exercise the
    start-up of the three-deep record ring; the dense symmetric weights the 128-entry record."""
    r = np.random.default_rng(11)
    Qd = r.uniform(-1, 1, (12, 12)); Qd = Qd @ Qd.T + 12 * np.eye(12)
    Rd = r.uniform(-0.3, 0.3, (4, 4)); Rd = Rd + Rd.T + 2 * np.eye(4)
    for B, n, dense, prec in [(33, 60, False, "f64"), (5, 1, False, "f64"), (7, 2, False, "f64"), (9, 3, False, "f64"),
                              (6, 4, True, "f64"), (20, 37, True, "f64"), (16, 50, False, "f32")]:
        cfg = pb.config2(B=B, N=n, seed=3)
        if dense:
            cfg["Q"], cfg["R"] = Qd, Rd
        two = capi_diag().from_config(cfg, precision=prec, force_general=3)  # k_backward2 (diagnostics build)
        one = capi.from_config(cfg, precision=prec, force_general=2)
        trajs = two.forward_sim(cfg["init"], np.zeros((B, n, 52)), 1.0)
        g2, t2 = two.backwards_pass(trajs)
        g1, t1 = one.backwards_pass(trajs)
        # (the two kernels read records in different placements, written by two instantiations of k_linearize, which the
        # compiler contracts into fused multiply-adds differently: the records agree to rounding, not bit for bit, and the
        # recursion carries that to about 1e-11 of the largest gain -- each is as far from the oracle as from the other;
        # in the mixed mode the records are computed in fp32)
        rt = 1e-10 if prec == "f64" else 1e-6
        np.testing.assert_allclose(t2, t1, rtol=rt, atol=1e-300)
        np.testing.assert_allclose(g2, g1, rtol=1e-9 if prec == "f64" else 1e-3,
                                   atol=(5e-11 if prec == "f64" else 1e-4) * max(np.abs(g1).max(), 1e-300))
        # whole solves.  In the mixed mode the default tolerances (1e-12) are below what fp32 records can resolve, so the two
        # placements' rounding decides where a solve stops: there the kernels that share a placement (k_backward2, k_backward4)
        # are held to each other, and to the one-wavefront kernel by the final cost only
        o2, o1 = two.solve_batch(cfg["init"]), one.solve_batch(cfg["init"])
        if prec == "f64":
            np.testing.assert_array_equal(o2["status"], o1["status"])
            np.testing.assert_array_equal(o2["iters"], o1["iters"])
            np.testing.assert_array_equal(o2["n_fwd"], o1["n_fwd"])
        np.testing.assert_allclose(o2["cost"], o1["cost"], rtol=1e-9 if prec == "f64" else 1e-4)
        # k_backward4: one gradient wavefront for four trajectories (blocks of four: B is not always a multiple); it reads the
        # records k_backward2 reads
        # ... and its fused form (matrix and gradient recursion in one wavefront: the one-wavefront kernel's gradient order)
        fused = capi.from_config(cfg, precision=prec, force_general=5)
        g5, t5 = fused.backwards_pass(trajs)
        np.testing.assert_allclose(t5, t2, rtol=1e-11 if prec == "f64" else 1e-6, atol=1e-300)
        np.testing.assert_allclose(g5, g2, rtol=1e-9 if prec == "f64" else 1e-5, atol=(1e-11 if prec == "f64" else 1e-5) * max(np.abs(g2).max(), 1e-300))
        o5 = fused.solve_batch(cfg["init"])
        if prec == "f64":
            np.testing.assert_array_equal(o5["status"], o2["status"])
            np.testing.assert_array_equal(o5["iters"], o2["iters"])
            np.testing.assert_array_equal(o5["n_fwd"], o2["n_fwd"])
        np.testing.assert_allclose(o5["cost"], o2["cost"], rtol=1e-9 if prec == "f64" else 1e-4)
        four = capi.from_config(cfg, precision=prec, force_general=4)
        g4, t4 = four.backwards_pass(trajs)
        np.testing.assert_allclose(t4, t2, rtol=1e-11, atol=1e-300)
        # (since round 6 the six-wavefront form performs the FUSED form's arithmetic, not k_backward2's: equal to the fused form bit for
        # bit, and as far from k_backward2 as the fused form is)
        np.testing.assert_array_equal(g4, g5)
        np.testing.assert_array_equal(t4, t5)
        np.testing.assert_allclose(g4, g2, rtol=1e-9 if prec == "f64" else 1e-5, atol=(1e-11 if prec == "f64" else 1e-5) * max(np.abs(g2).max(), 1e-300))
        np.testing.assert_allclose(t4, t1, rtol=1e-11 if prec == "f64" else 1e-6, atol=1e-300)
        o4 = four.solve_batch(cfg["init"])
        if prec == "f64":
            np.testing.assert_array_equal(o4["status"], o2["status"])
            np.testing.assert_array_equal(o4["iters"], o2["iters"])
            np.testing.assert_array_equal(o4["n_fwd"], o2["n_fwd"])
        np.testing.assert_allclose(o4["cost"], o2["cost"], rtol=1e-9 if prec == "f64" else 1e-4)


def test_backward_pass_bits_do_not_depend_on_the_batch_size():
    """Round 6: the six-wavefront k_backward4 (what a call with more than 4096 trajectories in flight takes; forced
    here at every size with force_general = 4) and the fused form (up to 4096; force_general = 5) perform the SAME arithmetic: H accumulated
    in one order, M^T V_x as three multiply-adds per quarter and (p0 + p1) + (p2 + p3), K^T Q_u as one chain, and Q_uu -- which the
    gradient wavefront of the six-wavefront form now builds and factors itself from the previous knot's V_xx -- the very bits of the
    accumulator tile's rows 12..15 (profiles/microbench/mfma_arith.hip).  Gains and expected-reduction terms are compared BIT FOR BIT:
    both register budgets of the six-wavefront form (batches on either side of 4096), horizons 1..4 (ring start-up, loop remainders),
    dense symmetric weights, the mixed mode, Levenberg-Marquardt mu on the diagonal of Q_uu, and whole solves on one rollout kernel."""
    r = np.random.default_rng(21)
    Qd = r.uniform(-1, 1, (12, 12)); Qd = Qd @ Qd.T + 12 * np.eye(12)
    Rd = r.uniform(-0.3, 0.3, (4, 4)); Rd = Rd + Rd.T + 2 * np.eye(4)
    for B, n, dense, prec in [(33, 60, False, "f64"), (5, 1, False, "f64"), (7, 2, False, "f64"), (9, 3, False, "f64"), (6, 4, True, "f64"),
                              (20, 37, True, "f64"), (16, 50, False, "f32"), (4100, 21, False, "f64"), (4099, 10, True, "f64"),
                              (64, 100, False, "f64"), (12, 150, False, "f64")]:
        cfg = pb.config2(B=B, N=n, seed=3 + n)
        if dense:
            cfg["Q"], cfg["R"] = Qd, Rd
        fused = capi.from_config(cfg, precision=prec, force_general=5)
        trajs = fused.forward_sim(cfg["init"], np.zeros((B, n, 52)), 1.0)
        g5, t5 = fused.backwards_pass(trajs)
        assert np.isfinite(g5).all() and (n < 3 or np.abs(g5).max() > 0)  # (the last knot's gains are zero: V = 0 behind it)
        # 4: the six-wavefront form as a call takes it (who factors Q_uu goes by the running trajectories); 7: the gradient wavefront
        # factors at every launch; 8: the matrix wavefronts do
        for fg in (4, 7, 8):
            g6, t6 = capi.from_config(cfg, precision=prec, force_general=fg).backwards_pass(trajs)
            np.testing.assert_array_equal(t6, t5, err_msg=f"terms force_general={fg} B={B} n={n} dense={dense} {prec}")
            np.testing.assert_array_equal(g6, g5, err_msg=f"gains force_general={fg} B={B} n={n} dense={dense} {prec}")
    # the automatic choice on either side of 4096: the first 64 of 4100 problems as part of that batch and alone
    cfg = pb.config2(B=4100, N=30, seed=8)
    trajs = capi.from_config(cfg).forward_sim(cfg["init"], np.zeros((4100, 30, 52)), 1.0)
    gw, tw = capi.from_config(cfg).backwards_pass(trajs)
    gs, ts = capi.from_config(cfg).backwards_pass(trajs[:64])
    np.testing.assert_array_equal(gw[:64], gs)
    np.testing.assert_array_equal(tw[:64], ts)
    # whole solves, one rollout kernel (k_rollout16 forced: single_wave_rollout = 3), with restarts (mu on the diagonal of Q_uu)
    cfg = pb.config2(B=96, N=40, seed=9)
    cfg["options"] = dict(cfg["options"], ls_max_iters=1)
    outs = []
    for fg in (5, 4, 7, 8):
        s = capi.from_config(cfg, force_general=fg, single_wave_rollout=3)
        s.set_regularisation(1.0, 4.0, 1e6)
        outs.append(s.solve_batch(cfg["init"]))
    assert (outs[0]["n_bwd"] > outs[0]["iters"] + 1).any()  # some problem restarted
    for o in outs[1:]:
        for k in ("traj", "cost", "status", "iters", "n_bwd", "n_fwd"):
            np.testing.assert_array_equal(outs[0][k], o[k], err_msg=k)


def test_sub_batches_on_their_own_streams_give_identical_results():
    """qilqr_device_config.streams: the batch is cut into contiguous tile ranges whose rounds run on separate
    HIP streams.  Trajectories are independent, so every output is bit-identical to the one-stream solve
    (ragged last tile, per-problem desired trajectories, cost history included)."""
    for B, n, streams in [(300, 40, 3), (130, 25, 2), (1000, 30, 8)]:
        cfg = pb.config2(B=B, N=n, seed=5)
        cfg["options"] = dict(cfg["options"], populate_debug=True)
        one = capi.from_config(cfg, sync_every=2, streams=1)
        many = capi.from_config(cfg, sync_every=2, streams=streams)
        r = np.random.default_rng(B)
        des = np.repeat(cfg["desired"][None, :n], B, 0)
        des[:, :, 1:4] += 0.2 * r.standard_normal((B, 1, 3))
        for desired_batch in (None, des):
            a = one.solve_batch(cfg["init"], desired_batch)
            b = many.solve_batch(cfg["init"], desired_batch)
            for k in ("status", "iters", "n_bwd", "n_fwd", "cost", "traj"):
                np.testing.assert_array_equal(a[k], b[k], err_msg=k)
            np.testing.assert_array_equal(np.nan_to_num(one.cost_history(B)), np.nan_to_num(many.cost_history(B)))


def test_rollout_kernels_agree():
    """The three rollout kernels on the same inputs.  k_rollout3 (pose wave + control wave + loader wave) performs the
    operations of k_rollout (one wave, a lane per trajectory): agreement to 1e-12 (the compiler may contract
    multiply-adds differently).  k_rollout16 (sixteen lanes per trajectory, rollout16.h) sums in other orders:
    1e-10, the per-pass tolerance.  Ragged batches (B not a multiple of 4 or 64), horizons 1 and 2, four different
    step sizes inside one block, and both storage precisions."""
    for B, n in [(70, 33), (5, 1), (64, 2), (3, 7)]:
        cfg = pb.config2(B=B, N=n, seed=7)
        r = np.random.default_rng(B)
        gains = 0.05 * r.uniform(-1, 1, (B, n, 52))
        alpha = 0.5 ** r.integers(0, 4, B)
        trajs = cfg["init"] + 0.0
        trajs[:, :, 8:14] += 0.3 * r.standard_normal((B, n, 6))
        one = capi.from_config(cfg, single_wave_rollout=1).forward_sim(trajs, gains, alpha)     # k_rollout
        three = capi.from_config(cfg, single_wave_rollout=2).forward_sim(trajs, gains, alpha)   # k_rollout3
        sixteen = capi.from_config(cfg, single_wave_rollout=3).forward_sim(trajs, gains, alpha)  # k_rollout16
        np.testing.assert_allclose(three, one, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(sixteen, one, rtol=1e-10, atol=1e-11)
        np.testing.assert_array_equal(sixteen[:, :, 0], trajs[:, :, 0])
        np.testing.assert_array_equal(sixteen[:, 0, 1:14], trajs[:, 0, 1:14])
        s32 = capi.from_config(cfg, single_wave_rollout=3, precision="f32").forward_sim(trajs, gains, alpha)
        np.testing.assert_allclose(s32, one, rtol=2e-5, atol=2e-5)
    cfg = pb.config2(B=96, N=40)
    outs = [capi.from_config(cfg, single_wave_rollout=k).solve_batch(cfg["init"]) for k in (1, 2, 3, 0)]
    for o in outs[1:]:
        np.testing.assert_allclose(o["traj"], outs[0]["traj"], atol=1e-8)
        np.testing.assert_array_equal(o["iters"], outs[0]["iters"])
        np.testing.assert_array_equal(o["n_fwd"], outs[0]["n_fwd"])


def test_rollout_fast_spins_through_the_three_ranges_of_exp():
    """The body-rate step's Exp (manif's SO3Tangent::exp, quadrotor_model.cc:47) by |dt omega|^2: up to 0.25 an eight-term series, up to 12 (round
    6, k_rollout16 only) a sixteen-term series, beyond the closed forms -- several ranges inside one wavefront, values either side of both
    boundaries, every rollout kernel against the oracle's forward_sim."""
    x2 = np.array([1e-12, 0.2, 0.249, 0.251, 0.26, 1.0, 3.0, 6.0, 9.0, 11.5, 11.99, 12.01, 13.0, 20.0, 30.0, 39.0, 0.0, 6.1, 2.0])
    B, n = len(x2), 14
    cfg = pb.config2(B=B, N=n, seed=3)
    r = np.random.default_rng(17)
    trajs = cfg["init"].copy()
    d = r.normal(size=(B, 3))
    trajs[:, 0, 11:14] = d / np.linalg.norm(d, axis=1)[:, None] * np.sqrt(x2)[:, None] / cfg["dt"]
    gains = np.zeros((B, n, 52))
    gains[:, :, :4] = 0.01 * r.uniform(-1, 1, (B, n, 4))   # a little feed-forward, no feedback: the spin persists over the horizon
    ref = oracle_for(cfg)
    want = np.stack([ref.forward_sim(trajs[b], gains[b], 1.0) for b in range(B)])
    w = np.linalg.norm(want[:, :, 11:14], axis=2) * cfg["dt"]
    assert (w[5:16] ** 2 > 0.25).all()   # (the fast rows stay beyond the eight-term series at every knot)
    for choice, tol in ((1, 1e-12), (2, 1e-12), (3, 1e-11)):
        got = capi.from_config(cfg, single_wave_rollout=choice).forward_sim(trajs, gains, 1.0)
        np.testing.assert_allclose(got, want, rtol=tol, atol=tol, err_msg=f"single_wave_rollout={choice}")


# ------------------------------------------------------------------ full solves
def test_config2_batch_matches_oracle():
    cfg = pb.config2(B=64)
    s = capi.from_config(cfg)
    out = s.solve_batch(cfg["init"])
    ref = oracle_for(cfg).solve_batch(cfg["init"], n_threads=8)
    np.testing.assert_array_equal(out["status"], ref["status"])
    np.testing.assert_array_equal(out["iters"], ref["iters"])
    np.testing.assert_array_equal(out["n_bwd"], ref["n_bwd"])
    np.testing.assert_array_equal(out["n_fwd"], ref["n_fwd"])
    np.testing.assert_allclose(out["cost"], ref["cost"], rtol=1e-9)
    np.testing.assert_allclose(out["traj"], ref["traj"], atol=1e-6)
    # and the first 8 against the committed fixture
    np.testing.assert_allclose(out["cost"][:8], G["cfg2_cost"], rtol=1e-9)
    np.testing.assert_allclose(out["traj"][:8], G["cfg2_traj"], atol=1e-6)
    np.testing.assert_array_equal(out["iters"][:8], G["cfg2_iters"])


def test_demo40_single_solve_matches_golden():
    # the reference demo as shipped (quadrotor_ilqr.py:256-306, 40 knots) through qilqr_solve
    cfg = pb.config1(4.0)
    s = capi.from_config(cfg)
    traj, info = s.solve(cfg["init"][0])
    meta = G["demo40_meta"]
    assert [info["status"], info["iters"]] == list(meta[:2])
    np.testing.assert_allclose(info["debug_costs"], G["demo40_cost_hist"], rtol=1e-8)
    np.testing.assert_allclose(info["cost"], G["demo40_cost_hist"][-1], rtol=1e-9)
    np.testing.assert_allclose(traj, G["demo40_traj"], atol=1e-6)
    np.testing.assert_array_equal(traj[:, 0], cfg["init"][0][:, 0])       # time_s passes through
    np.testing.assert_array_equal(traj[0, 1:14], cfg["init"][0][0, 1:14])  # knot 0 state is the input's


def test_demo100_config1():
    """BASELINE.json configs[0]: the demo at 100 knots.  This problem is chaotic IN THE REFERENCE
    ALGORITHM ITSELF: the desired roll reaches exactly pi (quadrotor_ilqr.py:101-106), the
    unchecked first full step (ilqr.hh:71-73) throws the rollout across the Log branch cut, and a
    1e-14 perturbation of the input moves the oracle's first-iteration cost by 4e-6 relative
    (tests/test_oracle_golden.py::test_demo100_is_chaotic).  No two floating-point implementations
    can agree to 1e-8 on it, so the bar here is: identical first backward pass (same inputs), the
    same exit path and iteration count, and the cost history within 1e-3."""
    cfg = pb.config1(10.0)
    s = capi.from_config(cfg)
    gains, terms = s.backwards_pass(cfg["init"])
    np.testing.assert_allclose(terms[0], G["demo100_terms0"], rtol=1e-10)
    # yaw is uncontrollable in the demo model (torque_to_thrust = 0): the corresponding gain entries
    # are rounding noise of size 1e-10 in both implementations, so compare against the gain scale
    np.testing.assert_allclose(gains[0], G["demo100_gains0"], rtol=1e-6, atol=1e-8 * np.abs(G["demo100_gains0"]).max())
    traj, info = s.solve(cfg["init"][0])
    assert [info["status"], info["iters"]] == list(G["demo100_meta"][:2])  # max_iters, 100
    np.testing.assert_allclose(info["debug_costs"], G["demo100_cost_hist"], rtol=1e-3)
    assert info["debug_costs"][-1] < info["debug_costs"][0] / 50
    np.testing.assert_array_equal(traj[:, 0], cfg["init"][0][:, 0])


def test_dense_coupled_weights_solve():
    """Full solves with dense (fully coupled, symmetric positive definite) Q and R.  Non-symmetric
    weights are covered at pass level (test_passes_match_oracle, dense=True): with them the
    reference's C_x = 2 dx^T Q J is not the cost gradient (cost.hh:51), its line search stalls on
    rounding noise, and a full solve has no reproducible answer in any implementation."""
    cfg = pb.config2(B=4, N=30)
    r = np.random.default_rng(21)
    A = r.uniform(-1, 1, (12, 12))
    D = np.sqrt(np.diag(pb.Q_DEMO))
    A4 = r.uniform(-1, 1, (4, 4))
    cfg = dict(cfg, Q=pb.Q_DEMO + 0.1 * (D[:, None] * (A + A.T) * D[None, :]) / 2, R=pb.R_DEMO + 0.1 * (A4 + A4.T))
    out = capi.from_config(cfg).solve_batch(cfg["init"])
    ref = oracle_for(cfg).solve_batch(cfg["init"])
    np.testing.assert_array_equal(out["status"], ref["status"])
    np.testing.assert_array_equal(out["iters"], ref["iters"])
    np.testing.assert_array_equal(out["n_fwd"], ref["n_fwd"])
    np.testing.assert_allclose(out["cost"], ref["cost"], rtol=1e-9)
    np.testing.assert_allclose(out["traj"], ref["traj"], atol=1e-6)


def test_non_symmetric_weights_whole_solves():
    """cost.hh:30-34 takes any dense Q, R; non-symmetric weights take the general kernel (k_backward<false>: the
    reference's own forms, Eigen's pivoted LDL^T).  VERDICT r05 missing 3 / item 4.
    (a) Where the iteration is well posed -- short horizons: ilqr.hh:133 amplifies Q's antisymmetric part by ~ 1.3 per knot
        (tests/test_oracle_nonsymmetric.py), and through its first iterations every decision sits far from its threshold -- WHOLE SOLVES are
        held to the oracle with the bars of the symmetric path: exit status, iteration, pass and rollout counts equal, cost history 1e-8,
        final cost 1e-9, trajectory 1e-6.  Q and R both non-symmetric (5 %).
    (b) At configs[1]'s horizon (bench.py's `reference_faithful.non_symmetric_Q` sample: 100 knots, Q + 0.05 triu) the reference algorithm
        itself has no answer: the first backward pass's gains are the amplified antisymmetric part, the first UNCHECKED step (ilqr.hh:71-73)
        takes the cost from 1e1..1e2 to 1e21..1e22, and every later decision is rounding of a rollout through those gains.  Held here to what
        is arithmetic-independent: the blow-up (the cost after the first step agrees with the oracle's to the few per cent by which the
        oracle's own two arithmetics differ), the final cost (nothing later moves it by more than a few per cent), and the exit classes -- never
        the expected-reduction convergence of a well-posed solve; mostly the exhausted search, like the oracle's fused-multiply-add build, whose
        arithmetic the device shares."""
    r7, r8 = np.random.default_rng(7), np.random.default_rng(8)
    U, Ur = np.triu(r7.uniform(-1, 1, (12, 12)), 1), np.triu(r8.uniform(-1, 1, (4, 4)), 1)
    for N, max_iters in ((12, 3), (20, 3), (30, 5)):
        cfg = pb.config2(B=24, N=N, seed=2)
        cfg["Q"], cfg["R"] = cfg["Q"] + 0.05 * U, cfg["R"] + 0.05 * Ur
        cfg["options"] = dict(cfg["options"], max_iters=max_iters, populate_debug=True)
        s = capi.from_config(cfg)
        assert "general" in s.describe(24)
        out = s.solve_batch(cfg["init"])
        o = oracle_for(cfg)
        ref = o.solve_batch(cfg["init"])
        for k in ("status", "iters", "n_bwd", "n_fwd"):
            np.testing.assert_array_equal(out[k], ref[k], err_msg=f"N={N} {k}")
        np.testing.assert_allclose(out["cost"], ref["cost"], rtol=1e-9)
        np.testing.assert_allclose(out["traj"], ref["traj"], atol=1e-6)
        hist = s.cost_history(24)
        for b in range(24):
            h = o.solve_decisions(cfg["init"][b])["cost_hist"]
            np.testing.assert_allclose(hist[b, :len(h)], h, rtol=1e-8)
    # (b) the bench's sample
    cfg = pb.config2(B=64, N=100, seed=2)
    cfg["Q"] = cfg["Q"] + 0.05 * np.triu(np.random.default_rng(7).uniform(-1, 1, (12, 12)), 1)
    cfg["options"] = dict(cfg["options"], populate_debug=True)
    s = capi.from_config(cfg)
    out = s.solve_batch(cfg["init"])
    hist = s.cost_history(64)
    P = oracle_for(cfg)
    F = orc.OracleSolver(orc.model_params(**cfg["model"]), cfg["Q"], cfg["R"], cfg["desired"], cfg["dt"], orc.options(**cfg["options"]),
                         library=orc.fast_library(native=False))
    rf = F.solve_batch(cfg["init"], n_threads=8)
    assert (out["status"] != 0).all() and np.sum(out["status"] == 3) >= 24 and np.sum(np.isin(out["status"], [1, 3])) >= 62, np.bincount(out["status"], minlength=4)
    assert np.sum(rf["status"] == 3) >= 24
    for b in range(64):
        c0 = P.cost_trajectory(cfg["init"][b])
        hp = P.solve_decisions(cfg["init"][b])["cost_hist"]
        assert hist[b, 0] > 1e15 * c0 and hp[0] > 1e15 * c0
        assert abs(hist[b, 0] - hp[0]) < 0.2 * hp[0], (b, hist[b, 0], hp[0])
        # it never comes back: no arithmetic repairs the solve.  (Not even monotone from there: with these gains Q_uu is indefinite, the
        # "expected reduction" of ilqr.hh:186 can be positive, and an accepted step may RAISE the cost.)
        assert out["cost"][b] > 1e15 * c0 and 0.1 < out["cost"][b] / hist[b, 0] < 10.0, (b, out["cost"][b], hist[b, 0])
        assert 0.02 < out["cost"][b] / rf["cost"][b] < 50.0, (b, out["cost"][b], rf["cost"][b])
    ratio = np.median(out["cost"] / rf["cost"])
    assert 0.8 < ratio < 1.25, ratio   # (typically the first step's cost to a few per cent, as between the oracle's own two arithmetics)
    # one pass on the first iterate is well defined in both: the gains against the oracle's.  (The recursion amplifies the two
    # implementations' rounding by the same 1.3 per knot: 1e-16 x 1.3^100 -- the bar is the long-horizon one, not the 1e-9 of a stable pass.)
    g, tm = s.backwards_pass(cfg["init"][:8])
    for b in range(8):
        go, to = P.backwards_pass(cfg["init"][b])
        assert np.max(np.abs(g[b] - go)) < 1e-4 * np.max(np.abs(go)), (b, np.max(np.abs(g[b] - go)) / np.max(np.abs(go)))
        np.testing.assert_allclose(g[b][60:], go[60:], rtol=0, atol=1e-9 * np.max(np.abs(go)))  # the last 40 knots: amplification still small


def test_batch_cost_history_matches_oracle():
    """ILQRDebug for batches: the cost after every completed forward pass"""
    cfg = pb.config2(B=12, N=60)
    cfg["options"] = dict(cfg["options"], populate_debug=True)
    s = capi.from_config(cfg)
    out = s.solve_batch(cfg["init"])
    hist = s.cost_history(12)
    ref = oracle_for(cfg)
    for b in range(12):
        r = ref.solve(cfg["init"][b])
        k = out["iters"][b]
        assert k == r["iters"] and np.isnan(hist[b, k:]).all()
        np.testing.assert_allclose(hist[b, :k], r["cost_hist"], rtol=1e-8)
        assert hist[b, k - 1] == out["cost"][b]


def test_per_problem_desired_trajectories():
    cfg = pb.config2(B=6, N=30)
    r = np.random.default_rng(5)
    des = np.broadcast_to(cfg["desired"], (6, 30, 18)).copy()
    des[:, :, 1:4] += r.uniform(-0.5, 0.5, (6, 1, 3))  # a different hover point per problem
    s = capi.from_config(cfg)
    out = s.solve_batch(cfg["init"], desired_batch=des)
    for b in range(6):
        ref = oracle_for(dict(cfg, desired=des[b])).solve(cfg["init"][b])
        assert out["status"][b] == ref["status"] and out["iters"][b] == ref["iters"]
        np.testing.assert_allclose(out["cost"][b], ref["cost"], rtol=1e-9)
        np.testing.assert_allclose(out["traj"][b], ref["traj"], atol=1e-6)


# ------------------------------------------------------------------ failure paths
def test_line_search_exhaustion():
    """ilqr.hh:191-193.  Made deterministic by asking for 10x the predicted reduction
    (desired_reduction_frac = 10): no step can satisfy the Armijo test, so iteration 1 exhausts its
    ls_max_iters trials in every implementation."""
    cfg = pb.config2(B=5, N=20)
    cfg["options"] = dict(cfg["options"], desired_reduction_frac=10.0, ls_max_iters=7)
    s = capi.from_config(cfg)
    with pytest.raises(RuntimeError, match=r"Reached maximum number of line search iterations, 7\n"):
        s.solve(cfg["init"][0])
    out = s.solve_batch(cfg["init"])
    ref = oracle_for(cfg).solve_batch(cfg["init"])
    assert (out["status"] == capi.STATUS_LINE_SEARCH_FAILED).all() and (ref["status"] == 3).all()
    np.testing.assert_array_equal(out["iters"], 1)       # iteration 0 is taken unconditionally
    np.testing.assert_array_equal(out["n_fwd"], 1 + 7)
    np.testing.assert_array_equal(out["n_bwd"], 2)
    # the trajectory handed back is the last accepted one (the reference throws instead)
    np.testing.assert_allclose(out["traj"], ref["traj"], atol=1e-9)
    np.testing.assert_allclose(out["cost"], ref["cost"], rtol=1e-10)
    # stand-alone line_search reports the same
    traj = s.forward_sim(cfg["init"], np.zeros((5, 20, 52)), 1.0)
    cost = s.cost_trajectory(traj)
    gains, terms = s.backwards_pass(traj)
    ls = s.line_search(traj, cost, gains, terms)
    assert (ls["status"] == capi.STATUS_LINE_SEARCH_FAILED).all()


def _restart_cfg(B=24, n=30, ls_max_iters=1, seed=7):
    cfg = pb.config2(B=B, N=n, seed=seed)
    cfg["init"][:, 0, 8:14] *= 4.0  # faster starts: more rejected full steps
    cfg["options"] = dict(cfg["options"], ls_max_iters=ls_max_iters)
    return cfg


@pytest.mark.parametrize("kernel", [0, 1, 2, 3, 4, 5])
def test_levenberg_marquardt_restarts_match_oracle(kernel):
    """qilqr_set_regularisation (an extension; the oracle states it, no reference
    behaviour to match): with one trial per line search most problems exhaust it and restart with mu on the
    diagonal of Q_uu.  Every backward kernel carries mu; all must follow the oracle's sequence of restarts,
    trials and accepted steps."""
    cfg = _restart_cfg()
    o = oracle_for(cfg)
    plain = o.solve_batch(cfg["init"], n_threads=8)
    assert (plain["status"] == 3).sum() >= 8
    o.set_regularisation(1.0, 4.0, 1e6)
    ref = o.solve_batch(cfg["init"], n_threads=8)
    assert (ref["status"] != 3).all() and (ref["n_bwd"] > ref["iters"] + 1).sum() >= 8
    s = (capi_diag() if kernel == 3 else capi).from_config(cfg, force_general=kernel)  # (3 = k_backward2: diagnostics build)
    off = s.solve_batch(cfg["init"])  # default: off, the reference's behaviour
    np.testing.assert_array_equal(off["status"], plain["status"])
    np.testing.assert_array_equal(off["n_bwd"], plain["n_bwd"])
    s.set_regularisation(1.0, 4.0, 1e6)
    out = s.solve_batch(cfg["init"])
    for k in ("status", "iters", "n_bwd", "n_fwd"):
        np.testing.assert_array_equal(out[k], ref[k], err_msg=k)
    np.testing.assert_allclose(out["cost"], ref["cost"], rtol=1e-9)
    np.testing.assert_allclose(out["traj"], ref["traj"], atol=1e-6)
    s.set_regularisation(0.0)
    again = s.solve_batch(cfg["init"])
    np.testing.assert_array_equal(again["status"], plain["status"])
    np.testing.assert_array_equal(again["traj"], off["traj"])


def test_restarts_past_mu_max_and_argument_checks():
    cfg = _restart_cfg(B=4, n=10, ls_max_iters=3)
    cfg["options"] = dict(cfg["options"], desired_reduction_frac=10.0)
    s = capi.from_config(cfg)
    s.set_regularisation(0.5, 2.0, 4.0)  # 0.5, 1, 2, 4: four restarts, then ilqr.hh:191-193
    out = s.solve_batch(cfg["init"])
    assert (out["status"] == capi.STATUS_LINE_SEARCH_FAILED).all()
    np.testing.assert_array_equal(out["iters"], 1)
    np.testing.assert_array_equal(out["n_bwd"], 2 + 4)
    np.testing.assert_array_equal(out["n_fwd"], 1 + 3 * 5)
    with pytest.raises(RuntimeError, match=r"Reached maximum number of line search iterations, 3\n"):
        s.solve(cfg["init"][0])
    for bad in [(1.0, 1.0, 10.0), (1.0, 2.0, 0.5), (-1.0, 2.0, 4.0), (float("nan"), 2.0, 4.0), (1e-300, 1.0000001, 1e300)]:
        with pytest.raises(ValueError):
            s.set_regularisation(*bad)


def test_restarts_with_sub_batches_on_their_own_streams():
    """The per-problem mu travels with the slice of the batch a stream owns: two and three parts give the
    one-stream results bit for bit, and the oracle's counts, with restarts on (ragged last tile)."""
    cfg = _restart_cfg(B=200, n=24, seed=9)
    o = oracle_for(cfg)
    o.set_regularisation(2.0, 8.0, 1e5)
    ref = o.solve_batch(cfg["init"], n_threads=8)
    assert (ref["n_bwd"] > ref["iters"] + 1).sum() >= 50
    outs = []
    for streams in (1, 2, 3):
        s = capi.from_config(cfg, streams=streams)
        s.set_regularisation(2.0, 8.0, 1e5)
        outs.append(s.solve_batch(cfg["init"]))
    assert_same_exit_paths(outs[0], ref, o, cfg["init"])
    np.testing.assert_allclose(outs[0]["cost"], ref["cost"], rtol=1e-9)
    for other in outs[1:]:
        for k in ("traj", "cost", "status", "iters", "n_bwd", "n_fwd"):
            np.testing.assert_array_equal(other[k], outs[0][k], err_msg=k)


def test_handles_in_flight_from_several_host_threads():
    """Per-handle re-entrancy (SURVEY.md section 8b "Threading"): three solver handles driven from three
    host threads at once -- the configuration bench.py's `serving` object measures -- give, bit for bit,
    what one handle gives solving the same batches one after the other."""
    import threading
    cfgs = [pb.config2(B=96, N=40, seed=21 + k) for k in range(3)]
    seq = capi.from_config(cfgs[0])
    want = [seq.solve_batch(c["init"]) for c in cfgs]  # same model / weights / desired trajectory in all three
    got = [None] * 3
    errs = []

    def drive(k):
        try:
            s = capi.from_config(cfgs[k])
            for _ in range(3):
                got[k] = s.solve_batch(cfgs[k]["init"])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=drive, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in range(3):
        for key in ("traj", "cost", "status", "iters", "n_bwd", "n_fwd"):
            np.testing.assert_array_equal(got[k][key], want[k][key], err_msg=f"{k} {key}")


def test_longer_than_desired_is_index_error():
    cfg = pb.config2(B=2, N=10)
    s = capi.from_config(cfg)
    longer = np.concatenate([cfg["init"], cfg["init"][:, -1:]], axis=1)
    with pytest.raises(IndexError):
        s.solve_batch(longer)
    with pytest.raises(IndexError):
        s.cost_trajectory(longer)


def test_unnormalised_initial_quaternion_is_value_error():
    cfg = pb.config2(B=2, N=10)
    s = capi.from_config(cfg)
    bad = cfg["init"].copy()
    bad[1, 3, 4:8] *= 1.01
    with pytest.raises(ValueError):
        s.solve_batch(bad)


def test_ragged_and_edge_sizes():
    # N = 1 (a single knot: no dynamics step at all), N = 2, B = 1, B not a multiple of 64; N = 3, 4, 6: horizons around the
    # four slots of k_backward4's operand ring (the loader's first tagged record, its first reuse of a slot)
    for B, n in [(1, 1), (3, 2), (65, 7), (130, 5), (2, 3), (5, 4), (7, 6)]:
        cfg = pb.config2(B=B, N=n, seed=9)
        s = capi.from_config(cfg)
        out = s.solve_batch(cfg["init"])
        ref = oracle_for(cfg).solve_batch(cfg["init"], n_threads=4)
        np.testing.assert_array_equal(out["status"], ref["status"])
        np.testing.assert_array_equal(out["iters"], ref["iters"])
        np.testing.assert_allclose(out["cost"], ref["cost"], rtol=1e-9, atol=1e-18)
        np.testing.assert_allclose(out["traj"], ref["traj"], atol=1e-6)


def test_max_iters_zero_returns_input():
    cfg = pb.config2(B=3, N=10)
    cfg["options"] = dict(cfg["options"], max_iters=0)
    out = capi.from_config(cfg).solve_batch(cfg["init"])
    np.testing.assert_array_equal(out["traj"], cfg["init"])
    assert (out["status"] == capi.STATUS_MAX_ITERS).all() and (out["iters"] == 0).all()


# ------------------------------------------------------------------ full size (BASELINE.json configs[1])
def test_config2_full_size_properties():
    cfg = pb.config2()  # B = 1024, N = 100
    s = capi.from_config(cfg)
    out = s.solve_batch(cfg["init"])
    assert np.isfinite(out["traj"]).all() and np.isfinite(out["cost"]).all()
    assert np.isin(out["status"], [0, 1]).all()
    # time column and knot-0 state pass through
    np.testing.assert_array_equal(out["traj"][:, :, 0], cfg["init"][:, :, 0])
    np.testing.assert_array_equal(out["traj"][:, 0, 1:14], cfg["init"][:, 0, 1:14])
    # the returned cost is the cost of the returned trajectory
    np.testing.assert_allclose(s.cost_trajectory(out["traj"]), out["cost"], rtol=1e-13)
    # problems are independent: any permutation of the batch gives the same per-problem bits
    perm = np.random.default_rng(0).permutation(1024)
    out_p = s.solve_batch(cfg["init"][perm])
    np.testing.assert_array_equal(out_p["traj"], out["traj"][perm])
    np.testing.assert_array_equal(out_p["iters"], out["iters"][perm])
    # A batch of one gives the same result as the same problem inside the batch (the same backward kernel up to 8192
    # trajectories since round 3; beyond, the one-wavefront kernel sums the twelve terms of M^T V_x in another order,
    # measured difference 2e-15).
    one = s.solve_batch(cfg["init"][17:18])
    assert one["iters"][0] == out["iters"][17] and one["status"][0] == out["status"][17]
    np.testing.assert_allclose(one["traj"][0], out["traj"][17], rtol=0, atol=1e-11)
    # with the kernel held fixed the bits are the same
    fixed = capi.from_config(cfg, force_general=4)
    np.testing.assert_array_equal(fixed.solve_batch(cfg["init"][17:18])["traj"][0], fixed.solve_batch(cfg["init"])["traj"][17])
    # warm start from the optimum is a fixed point: at most two more iterations, same cost
    again = s.solve_batch(out["traj"])
    assert (again["iters"] <= 2).all()
    np.testing.assert_allclose(again["cost"], out["cost"], rtol=1e-9)
    # a sample against the oracle
    idx = np.arange(0, 1024, 32)
    ref = oracle_for(cfg).solve_batch(cfg["init"][idx], n_threads=8)
    np.testing.assert_array_equal(out["iters"][idx], ref["iters"])
    np.testing.assert_allclose(out["cost"][idx], ref["cost"], rtol=1e-9)
    np.testing.assert_allclose(out["traj"][idx], ref["traj"], atol=1e-6)


# ------------------------------------------------------------------ BASELINE.json configs[4] at reduced size
def test_config5_long_horizon_stress_reduced():
    """Half the batch well posed (model A hover, parity with the oracle), half the demo's box-climb at a
    long horizon with random starts (divergent first step, heavy back-tracking, max-iteration / line
    search exits).  The second half is chaotic in the reference algorithm itself, so it is graded on
    the exit class and on internal consistency, not on digits."""
    a, b = pb.config5(B=16, N=150)
    sa, sb = capi.from_config(a), capi.from_config(b)
    oa, ob = sa.solve_batch(a["init"]), sb.solve_batch(b["init"])
    ra, rb = oracle_for(a).solve_batch(a["init"], n_threads=8), oracle_for(b).solve_batch(b["init"], n_threads=8)
    np.testing.assert_array_equal(oa["status"], ra["status"])
    np.testing.assert_array_equal(oa["iters"], ra["iters"])
    observed("configs[4] reduced, half A 16 x 150 vs oracle (reference form)", oa, ra)
    np.testing.assert_allclose(oa["cost"], ra["cost"], rtol=1e-9)
    np.testing.assert_allclose(oa["traj"], ra["traj"], atol=1e-5)  # 150 knots: the REFERENCE form's own drift (7e-3 in its gains there)
    # ... and against the oracle's symmetrised form of the recursion, which does not drift: the stated fp64 bars
    oracle1 = oracle_for(a, recursion=1)
    ra1 = oracle1.solve_batch(a["init"], n_threads=8)
    assert_same_exit_paths(oa, ra1, oracle1, a["init"], label="configs[4] reduced, 150 knots")
    observed("configs[4] reduced, half A 16 x 150 vs oracle (symmetrised)", oa, ra1)
    np.testing.assert_allclose(oa["cost"], ra1["cost"], rtol=1e-9)
    np.testing.assert_allclose(oa["traj"], ra1["traj"], atol=1e-6)
    assert np.isin(rb["status"], [2, 3]).all()          # the oracle does not converge on these either
    assert np.isin(ob["status"], [2, 3]).all()
    assert np.isfinite(ob["traj"]).all() and np.isfinite(ob["cost"]).all()
    assert (ob["n_fwd"] > ob["iters"]).all()            # back-tracking happened
    np.testing.assert_allclose(sb.cost_trajectory(ob["traj"]), ob["cost"], rtol=1e-12)
    # both implementations end in the same cost regime (orders of magnitude below the first rollout)
    assert np.all(np.abs(np.log10(ob["cost"] / rb["cost"])) < 1.0)


# ------------------------------------------------------------------ BASELINE.json configs[2]: fp32
def test_long_horizon_instability_of_the_unsymmetrised_recursion():
    """A finding, pinned: V_xx = Q_xx - K^T Q_uu K without symmetrisation (ilqr.hh:133)
    is numerically unstable over long horizons.  At 200 knots (model A, hover) the fp64 reference
    algorithm's own feedback gains are rounding garbage (|K| ~ 1e3 instead of ~1e1) and its first
    unchecked rollout diverges (costs > 1e9).  The general backward kernel (force_general) fails the same
    way; the symmetric-weights kernel, which reuses the accumulator tile as the next operand and takes the
    right-hand sides from the symmetric accumulator, stays bounded and converges.  At 100 knots all three agree (other tests)."""
    cfg = pb.config3(B=16, N=200)
    ref = oracle_for(cfg)
    g_ref, _ = ref.backwards_pass(cfg["init"][1])
    g_100, _ = oracle_for(pb.config3(B=2, N=100)).backwards_pass(pb.config3(B=2, N=100)["init"][1])
    assert np.abs(g_100).max() < 50 and np.abs(g_ref).max() > 500
    r = ref.solve_batch(cfg["init"], n_threads=8)
    assert (r["cost"] > 1e9).all()
    sym = capi.from_config(cfg).solve_batch(cfg["init"])
    gen = capi.from_config(cfg, force_general=True).solve_batch(cfg["init"])
    # same failure class as the reference: no problem reaches the optimum -- costs two to nine orders of magnitude above the convergent
    # regime -- and the gains of one backward pass are several times the bounded recursion's.  Which problem lands where, and how far, is
    # rounding noise (that is the finding): the oracle's dense products leave every cost above 1e9; the general kernel's matrix-core products
    # (round 5: an accumulator carried alternately as V and V^T, no transposes) leave them between 2e4 and 3e11, |K| ~ 2.4e2 against ~ 1e3
    # (any change of a rounding anywhere re-draws them: round 6's sixteen-term series in the rollout took the smallest ratio from 60 to 14.6).
    assert (gen["cost"] > 8 * sym["cost"]).all() and (gen["cost"] > 100 * sym["cost"]).sum() >= 12 and np.median(gen["cost"]) > 1e5 and (gen["cost"] > 1e8).sum() >= 4
    g_gen, _ = capi.from_config(cfg, force_general=True).backwards_pass(cfg["init"][1:2])
    assert np.abs(g_gen).max() > 100
    assert np.isin(sym["status"], [0, 1]).all() and (sym["cost"] < 1e4).all()
    g_sym, t_sym = capi.from_config(cfg).backwards_pass(cfg["init"][1:2])
    assert np.abs(g_sym).max() < 50
    # what the symmetric kernels DO compute at this horizon has a comparand outside the library: the oracle's symmetrised
    # form of the same recursion (orc_set_recursion(1)) -- one pass at the per-pass bar, the solves at the fp64 bars
    oracle1 = oracle_for(cfg, recursion=1)
    g1, t1 = oracle1.backwards_pass(cfg["init"][1])
    np.testing.assert_allclose(g_sym[0], g1, rtol=0, atol=1e-9 * np.abs(g1).max())
    np.testing.assert_allclose(t_sym[0], t1, rtol=1e-10)
    r1 = oracle1.solve_batch(cfg["init"], n_threads=8)
    assert_same_exit_paths(sym, r1, oracle1, cfg["init"], label="200 knots, symmetric kernels")
    observed("configs[2] family fp64 16 x 200 vs oracle (symmetrised)", sym, r1)
    np.testing.assert_allclose(sym["cost"], r1["cost"], rtol=1e-9)
    np.testing.assert_allclose(sym["traj"], r1["traj"], atol=1e-6)


def test_config3_mixed_precision_reduced():
    """fp32 storage + fp32 rollout / linearisation, fp64 Riccati recursion and fp64 cost arithmetic.
    Stated fp32 bar: final cost within 1e-3 relative, trajectory within 1e-2, exit paths
    convergence exits.  Checked (a) against the fp64 ORACLE at 100 knots, where the reference recursion
    is stable, and (b) at the configuration's 200 knots against this library's fp64 mode (the reference
    itself is unstable there: previous test)."""
    cfg = pb.config3(B=64, N=100)
    s32 = capi.from_config(cfg, precision="f32")
    out = s32.solve_batch(cfg["init"])
    ref = oracle_for(cfg).solve_batch(cfg["init"], n_threads=8)
    assert np.isin(out["status"], [0, 1]).all() and np.isin(ref["status"], [0, 1]).all()
    np.testing.assert_allclose(out["cost"], ref["cost"], rtol=1e-3)
    np.testing.assert_allclose(out["traj"], ref["traj"], atol=1e-2)
    assert np.abs(out["iters"].astype(int) - ref["iters"]).max() <= 3
    cfg = pb.config3(B=64, N=200)
    s32, s64 = capi.from_config(cfg, precision="f32"), capi.from_config(cfg)
    o32, o64 = s32.solve_batch(cfg["init"]), s64.solve_batch(cfg["init"])
    assert np.isin(o32["status"], [0, 1]).all() and np.isin(o64["status"], [0, 1]).all()
    np.testing.assert_allclose(o32["cost"], o64["cost"], rtol=1e-3)
    np.testing.assert_allclose(o32["traj"], o64["traj"], atol=1e-2)
    # (c) at 200 knots against the ORACLE's symmetrised form of the recursion: fp32 at the fp32 bar, fp64 at the fp64 bar
    oracle1 = oracle_for(cfg, recursion=1)
    r1 = oracle1.solve_batch(cfg["init"], n_threads=8)
    observed("configs[2] reduced fp32 64 x 200 vs oracle (symmetrised)", o32, r1)
    np.testing.assert_allclose(o32["cost"], r1["cost"], rtol=1e-3)
    np.testing.assert_allclose(o32["traj"][:, :, :14], r1["traj"][:, :, :14], atol=1e-2)
    assert_same_exit_paths(o64, r1, oracle1, cfg["init"], label="configs[2] reduced, fp64 at 200 knots")
    observed("configs[2] reduced fp64 64 x 200 vs oracle (symmetrised)", o64, r1)
    np.testing.assert_allclose(o64["cost"], r1["cost"], rtol=1e-9)
    np.testing.assert_allclose(o64["traj"], r1["traj"], atol=1e-6)
    # per-pass agreement of the fp32 kernels with the fp64 ones
    tr = s64.forward_sim(cfg["init"], np.zeros((64, 200, 52)), 1.0)
    np.testing.assert_allclose(s32.cost_trajectory(tr), s64.cost_trajectory(tr), rtol=2e-5)
    g32, t32 = s32.backwards_pass(tr)
    g64, t64 = s64.backwards_pass(tr)
    np.testing.assert_allclose(t32, t64, rtol=2e-3)
    np.testing.assert_allclose(g32, g64, rtol=0, atol=1e-3 * np.abs(g64).max())


# ------------------------------------------------------------------ randomised parity sweep
def test_mixed_precision_at_headline_size():
    """precision = 1 (fp32 storage and lane-local arithmetic, fp64 recursion and cost sums) on configs[1]'s shape,
    where the four-trajectory backward kernel is the one selected: final costs within 1e-5 relative of the fp64
    solve at tolerances fp32 can reach, trajectories within 1e-4."""
    cfg = pb.config2(B=1024, N=100, seed=2)
    cfg["options"] = dict(cfg["options"], rtol=1e-5, atol=1e-5)
    a = capi.from_config(cfg, precision="f32").solve_batch(cfg["init"])
    b = capi.from_config(cfg, precision="f64").solve_batch(cfg["init"])
    assert np.isin(a["status"], [0, 1]).all() and np.isin(b["status"], [0, 1]).all()
    np.testing.assert_allclose(a["cost"], b["cost"], rtol=1e-5)
    np.testing.assert_allclose(a["traj"], b["traj"], atol=1e-4)


def randomised_cfg(seed, restarts=False):
    """the problem of test_randomised_models_and_horizons_match_oracle(seed): (cfg, Levenberg-Marquardt schedule or None)"""
    r = np.random.default_rng(1000 + seed)
    A = r.uniform(-0.3, 0.3, (3, 3))
    model = dict(mass_kg=r.uniform(0.5, 3.0), inertia=A @ A.T + np.diag(r.uniform(0.5, 2.0, 3)),
                 arm_length_m=r.uniform(0.2, 1.2), torque_to_thrust_ratio_m=r.uniform(0.05, 0.5),
                 g_mpss=r.uniform(3.0, 12.0))
    n = int(r.integers(5, 90))
    dt = float(r.uniform(0.02, 0.12))
    B = int(r.integers(1, 40))
    Q = np.diag(np.concatenate([r.uniform(10, 200, 6), r.uniform(0.5, 5, 6)]))
    R = np.diag(r.uniform(0.5, 3.0, 4))
    if seed % 3 == 1:    # dense symmetric blocks, no pose x velocity coupling (block-diagonal record layout)
        for lo in (0, 6):
            G = r.uniform(-1, 1, (6, 6))
            Q[lo:lo + 6, lo:lo + 6] += (G @ G.T) * (3.0 if lo == 0 else 0.1)
        G = r.uniform(-0.3, 0.3, (4, 4))
        R = R + G @ G.T
    elif seed % 3 == 2:  # fully dense symmetric weights (pose x velocity block stored per knot)
        G = r.uniform(-1, 1, (12, 12))
        Q = Q + 0.3 * (G @ G.T)
        G = r.uniform(-0.3, 0.3, (4, 4))
        R = R + G @ G.T
    desired = pb.hover_desired(n, dt, model["mass_kg"] * model["g_mpss"] / 4.0)
    desired[:, 1:8] = orc.se3_exp(np.concatenate([r.uniform(-1, 1, 3), r.uniform(-0.3, 0.3, 3)]))
    init = pb.random_start_batch(np.arange(B), desired, 77 + seed, pos_m=0.8, ang_rad=0.6, vel_sigma=0.4)
    opts = dict(step_update=float(r.choice([0.5, 0.3, 0.7])), desired_reduction_frac=float(r.choice([0.5, 0.1, 0.01])),
                ls_max_iters=int(r.integers(5, 40)), rtol=1e-10, atol=1e-10, max_iters=int(r.integers(3, 60)),
                populate_debug=False)
    reg = None
    if restarts:  # few trials per line search, so that searches are exhausted, and Levenberg-Marquardt restarts on
        opts["ls_max_iters"] = int(r.integers(1, 4))
        reg = (float(r.choice([0.1, 1.0, 10.0])), float(r.choice([2.0, 4.0, 10.0])), float(r.choice([1e3, 1e6])))
    cfg = dict(model=model, Q=Q, R=R, dt=dt, options=opts, desired=desired, init=init)
    return cfg, reg


@pytest.mark.parametrize("seed", range(12))
def test_randomised_models_and_horizons_match_oracle(seed, restarts=False, persistent=0):
    """Random physical parameters (mass, SPD inertia, arm, rotor torque ratio, gravity), time step,
    horizon, weights (diagonal / block-diagonal dense / fully dense symmetric, by seed) and option values;
    random SE(3) starts towards a random hover pose."""
    cfg, reg = randomised_cfg(seed, restarts)
    init = cfg["init"]
    s, o = (capi_diag() if persistent == 1 else capi).from_config(cfg, persistent=persistent), oracle_for(cfg)
    if reg:
        s.set_regularisation(*reg)
        o.set_regularisation(*reg)
    out = s.solve_batch(init)
    ref = o.solve_batch(init, n_threads=8)
    np.testing.assert_array_equal(out["status"], ref["status"])
    np.testing.assert_array_equal(out["iters"], ref["iters"])
    np.testing.assert_array_equal(out["n_bwd"], ref["n_bwd"])
    np.testing.assert_array_equal(out["n_fwd"], ref["n_fwd"])
    observed(f"randomised family seed {seed} restarts {restarts} persistent {persistent}", out, ref)
    np.testing.assert_allclose(out["cost"], ref["cost"], rtol=1e-9)
    np.testing.assert_allclose(out["traj"], ref["traj"], atol=1e-6)


@pytest.mark.parametrize("seed", range(100, 108))
def test_randomised_restarts_match_oracle(seed):
    """The randomised case above with one to three trials per line search and Levenberg-Marquardt restarts
    (random mu_init, factor, mu_max) on both sides: same sequence of restarts, trials and accepted steps."""
    test_randomised_models_and_horizons_match_oracle(seed, restarts=True)


# ------------------------------------------------------------------ the persistent solve (k_solve4: one launch per batch)
@pytest.mark.parametrize("seed,restarts", [(s, False) for s in range(12)] + [(s, True) for s in range(100, 106)])
def test_persistent_solve_matches_oracle(seed, restarts):
    """The randomised cases above (models, horizons 5..90, batches 1..40 -- ragged groups of four --, dense and diagonal
    symmetric weights, few-trial line searches with Levenberg-Marquardt restarts) through k_solve4: same statuses,
    iteration and pass counts, costs and trajectories as the oracle."""
    test_randomised_models_and_horizons_match_oracle(seed, restarts=restarts, persistent=1)


def test_persistent_solve_is_one_launch_and_matches_the_rounds():
    """persistent = 1 really takes k_solve4 (one launch per batch solve, counted by the profile), in both storage precisions,
    with per-problem desired trajectories and the cost history; against the rounds on the same inputs: exit paths equal up
    to threshold flips (the two paths inline the same arithmetic into different kernels), costs and trajectories agree."""
    B, n = 203, 60
    cfg = pb.config2(B=B, N=n, seed=9)
    r = np.random.default_rng(5)
    desired_batch = np.repeat(cfg["desired"][None], B, axis=0)
    desired_batch[:, :, 1:4] += r.uniform(-0.2, 0.2, (B, 1, 3))
    for precision, tol in (("f64", 1e-6), ("f32", 2e-3)):
        c = dict(cfg, options=dict(cfg["options"], populate_debug=True))  # (the cost history)
        if precision == "f32":
            c["options"] = dict(c["options"], rtol=1e-5, atol=1e-5)
        pers = capi_diag().from_config(c, persistent=1, profile=1, precision=precision)
        rnds = capi.from_config(c, persistent=2, precision=precision)
        for desired in (None, desired_batch):
            pers.profile_reset()
            a = pers.solve_batch(cfg["init"], desired)
            p = pers.profile_get()
            assert p["solve_launches"] == 1 and p["backward_launches"] == 0 and p["rollout_launches"] == 0, p
            b = rnds.solve_batch(cfg["init"], desired)
            assert np.isin(a["status"], [0, 1]).all()
            if precision == "f64":
                # two paths of the library: where their counts differ, each must be the oracle's or one flipped comparison away
                differ = np.zeros(B, dtype=bool)
                for k in ("status", "iters", "n_bwd", "n_fwd"):
                    differ |= (a[k] != b[k])
                for i in np.nonzero(differ)[0]:
                    ci = c if desired is None else dict(c, desired=desired[i])
                    r = oracle_for(ci).solve_decisions(cfg["init"][i])
                    for name, res in (("k_solve4", a), ("rounds", b)):
                        got = tuple(int(res[k][i]) for k in ("status", "iters", "n_bwd", "n_fwd"))
                        d = explain(got, r)
                        if d is not None:
                            print(describe(int(i), got, r, d, name))
            np.testing.assert_allclose(a["cost"], b["cost"], rtol=1e-6 if precision == "f64" else 1e-3)
            np.testing.assert_allclose(a["traj"], b["traj"], atol=tol)
        if precision == "f64":
            ha, hb = pers.cost_history(B), rnds.cost_history(B)
            same = (a["iters"] == b["iters"])
            np.testing.assert_allclose(np.nan_to_num(ha[same]), np.nan_to_num(hb[same]), rtol=1e-9)


def test_diagonal_weights_path_gives_the_same_bits():
    """Q exactly diagonal (configs[1..4], the reference's demo and tests) takes the instantiation of k_linearize whose cost half
    scales rows instead of multiplying by Q (linearize_cost<3>); qilqr_device_config.dense_weights = 1 keeps the
    block-diagonal one.  The sums of the latter add exact zeros to the products the former keeps: whole solves agree bit
    for bit.  (fp64 solvers only: the mixed mode keeps the block-diagonal instantiation, and the switch changes nothing there.)"""
    import os
    cfg = pb.config2(B=96, N=70, seed=11)
    for prec in ("f64", "f32"):
        c = cfg if prec == "f64" else dict(cfg, options=dict(cfg["options"], rtol=1e-5, atol=1e-5))
        fast = capi.from_config(c, precision=prec).solve_batch(c["init"])
        plain = capi.from_config(c, precision=prec, dense_weights=1).solve_batch(c["init"])
        for k in ("status", "iters", "n_bwd", "n_fwd", "cost", "traj"):
            np.testing.assert_array_equal(fast[k], plain[k], err_msg=f"{prec} {k}")


def test_roctx_ranges_change_nothing():
    """qilqr_device_config.profile bit 16: roctx ranges around the call, every round and every sub-batch stream's share of a round (SURVEY.md
    section 5; for `rocprofv3 --marker-trace --kernel-trace`).  libroctx64 is bound with dlopen at first use; the results are those of the
    unmarked solve, on one stream and on sub-batch streams."""
    for B, streams in ((40, 0), (300, 3)):
        cfg = pb.config2(B=B, N=30, seed=6)
        a = capi.from_config(cfg, streams=streams).solve_batch(cfg["init"])
        b = capi.from_config(cfg, streams=streams, profile=0x10000).solve_batch(cfg["init"])
        for k in ("traj", "cost", "status", "iters", "n_bwd", "n_fwd"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_retired_kernel_choice_is_refused_by_name():
    """force_general = 6 (the fused k_backward4 with a block barrier per knot, round 3's A/B partner of the barrier-free form)
    was retired in round 4; the library says so instead of silently taking another kernel."""
    cfg = pb.config2(B=4, N=8)
    with pytest.raises(TypeError, match="retired in round 4"):
        capi.from_config(cfg, force_general=6)
