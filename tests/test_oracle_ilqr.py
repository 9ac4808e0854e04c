"""Oracle vs the reference's ilqr_test.cc fixture (KA1-KA6 of SURVEY.md section 8c).

Fixture (ilqr_test.cc:68-100): N=3, dt=0.1, m=1, I=eye, arm=1, ttr=1, g=0, Q=I12, R=I4,
desired = identity trajectory, LS{0.5,0.5,10}, rtol=atol=1e-12, max_iters=100.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from quadrotorilqr_amd import problems as pb

N, DT = 3, 0.1


@pytest.fixture(scope="module")
def fx():
    cur = pb.identity_trajectory(N, DT)
    mp = orc.model_params(1.0, np.eye(3), 1.0, 1.0, 0.0)
    opt = orc.options(0.5, 0.5, 10, 1e-12, 1e-12, 100)
    s = orc.OracleSolver(mp, np.eye(12), np.eye(4), cur, DT, opt)
    gains = orc.kK_to_gains(np.ones((N, 4)), np.zeros((N, 4, 12)))
    return dict(s=s, cur=cur, gains=gains)


def traj_close(a, b, tol=1e-6):
    """ilqr_test.cc:38-64: |log(a^-1 b)| < tol, velocities and controls isApprox or isZero(tol)"""
    assert a.shape == b.shape
    for pa, pb_ in zip(a, b):
        rel = orc.se3_compose(orc.se3_inverse(pa[1:8]), pb_[1:8])
        assert np.linalg.norm(orc.se3_log(rel)) < tol
        np.testing.assert_allclose(pa[8:14], pb_[8:14], atol=tol, rtol=1e-12)
        np.testing.assert_allclose(pa[14:18], pb_[14:18], atol=tol, rtol=1e-12)


# ---- ilqr_test.cc:102-126 (KA1)
def test_forward_sim_generates_correct_trajectory(fx):
    new = fx["s"].forward_sim(fx["cur"], fx["gains"])
    acc = 4.0
    exp = pb.identity_trajectory(N, DT)
    exp[:, 14:18] = 1.0
    exp[1, 10] = DT * acc
    exp[2, 3] = DT * DT * acc
    exp[2, 10] = 2 * DT * acc
    traj_close(new, exp)
    np.testing.assert_array_equal(new[:, 0], fx["cur"][:, 0])  # time passes through, ilqr.hh:164


# ---- ilqr_test.cc:128-141 (KA2): EXPECT_DOUBLE_EQ = within 4 ULP
def test_cost_trajectory_known_answer(fx):
    new = fx["s"].forward_sim(fx["cur"], fx["gains"])
    cost = fx["s"].cost_trajectory(new)
    acc = 4.0
    expected = (DT * acc) ** 2.0 + (DT * DT * acc) ** 2.0 + (2.0 * DT * acc) ** 2.0 + 3 * 4
    assert abs(cost - expected) <= 4 * np.spacing(expected)
    assert abs(cost - 12.8016) < 1e-12


# ---- ilqr_test.cc:143-153 (KA3)
def test_backward_pass_zero_update_if_zero_gradient(fx):
    gains, terms = fx["s"].backwards_pass(fx["cur"])
    assert gains.shape == (N, 52)
    assert terms[0] == 0.0 and terms[1] == 0.0
    k, _ = orc.gains_to_kK(gains)
    assert np.all(k == 0.0)


# ---- ilqr_test.cc:155-164 (KA4)
def test_backward_pass_expected_reduction_negative(fx):
    new = fx["s"].forward_sim(fx["cur"], fx["gains"])
    _, terms = fx["s"].backwards_pass(new)
    assert terms[0] < 0.0
    assert abs(terms[0] - (-25.6032)) < 1e-9  # SURVEY.md scratch value; = -2 * cost for this fixture


# ---- ilqr_test.cc:166-177 (KA5)
def test_line_search_finds_step_that_reduces_cost(fx):
    s = fx["s"]
    traj = s.forward_sim(fx["cur"], fx["gains"])
    cost = s.cost_trajectory(traj)
    gains, terms = s.backwards_pass(traj)
    ls = s.line_search(traj, cost, gains, terms)
    assert ls["status"] == 0
    dj = ls["step"] * terms[0] + ls["step"] ** 2 * terms[1] / 2.0
    assert ls["cost"] - cost < 0.5 * dj
    assert ls["cost"] == s.cost_trajectory(ls["traj"])


# ---- ilqr_test.cc:179-190 (KA6)
def test_solve_finds_optimal_trajectory(fx):
    k = np.ones((N, 4))
    k[:, 0] *= 100
    k[:, 2] *= 100
    init = fx["s"].forward_sim(fx["cur"], orc.kK_to_gains(k, np.zeros((N, 4, 12))))
    out = fx["s"].solve(init, debug=True)
    traj_close(fx["cur"], out["traj"], 1e-6)
    assert out["cost"] < 1e-20
    assert out["status"] in (orc.STATUS_CONVERGED_EXPECTED, orc.STATUS_CONVERGED)
    assert len(out["cost_hist"]) == out["iters"] == out["n_fwd"]
    np.testing.assert_array_equal(out["debug_trajs"][-1], out["traj"])  # returns the last accepted rollout


# ---- structure of the last knot: no terminal cost => K_{N-1}=0, k_{N-1}=-(u-u_d)
def test_last_knot_gains(fx):
    new = fx["s"].forward_sim(fx["cur"], fx["gains"])
    gains, _ = fx["s"].backwards_pass(new)
    k, K = orc.gains_to_kK(gains)
    np.testing.assert_allclose(K[-1], 0.0, atol=1e-15)
    np.testing.assert_allclose(k[-1], -(new[-1, 14:18] - 0.0), atol=1e-15)


# ---- cost.hh:39-40: .at(i) throws when the trajectory is longer than the desired one
def test_longer_than_desired_is_index_error(fx):
    with pytest.raises(IndexError):
        fx["s"].cost_trajectory(pb.identity_trajectory(N + 1, DT))
    with pytest.raises(IndexError):
        fx["s"].solve(pb.identity_trajectory(N + 1, DT))


# ---- ilqr.hh:191-193: exhaustion -> status LINE_SEARCH_FAILED (the reference throws)
def test_line_search_exhaustion_status():
    desired = pb.box_climb_desired(20.0)  # N=200 demo: exhausts at iteration 1
    mp = orc.model_params(**pb.MODEL_D)
    s = orc.OracleSolver(mp, pb.Q_DEMO, pb.R_DEMO, desired, pb.DT_DEMO, orc.options(**pb.OPTIONS_DEMO))
    out = s.solve(desired)
    assert out["status"] == orc.STATUS_LINE_SEARCH_FAILED
    # the failing iteration contributes exactly ls_max_iters rejected trials and no debug entry
    assert out["n_bwd"] == out["iters"] + 1
    assert out["n_fwd"] >= 100 + out["iters"] and len(out["cost_hist"]) == out["iters"]


# ---- Eigen pivoted LDLT restatement against numpy
def test_ldlt4_solve_matches_numpy():
    r = np.random.default_rng(3)
    for _ in range(20):
        A = r.uniform(-1, 1, (4, 4))
        A = A @ A.T + 0.1 * np.eye(4)
        B = r.uniform(-1, 1, (4, 13))
        np.testing.assert_allclose(orc.ldlt4_solve(A, B), np.linalg.solve(A, B), rtol=1e-10, atol=1e-12)
    A = np.diag([1.0, -2.0, 3.0, 0.5])  # indefinite: LDLT still solves it (Quu not PD is not detected)
    A[0, 1] = A[1, 0] = 0.3
    b = np.array([1.0, 2.0, 3.0, 4.0])
    np.testing.assert_allclose(orc.ldlt4_solve(A, b), np.linalg.solve(A, b), rtol=1e-12)


# ---- Levenberg-Marquardt restarts: an EXTENSION, not reference behaviour.
# These tests pin the oracle's statement of it, which the GPU path is compared with.
def _hover_cfg(B=24, n=30, ls_max_iters=1, seed=7):
    cfg = pb.config2(B=B, N=n, seed=seed)
    cfg["init"][:, 0, 8:14] *= 4.0  # faster starts: more rejected full steps
    cfg["options"] = dict(cfg["options"], ls_max_iters=ls_max_iters)
    return cfg


def _oracle(cfg):
    return orc.OracleSolver(orc.model_params(**cfg["model"]), cfg["Q"], cfg["R"], cfg["desired"], cfg["dt"],
                            orc.options(**cfg["options"]))


def test_regularised_backward_pass_is_the_lqr_step_of_a_heavier_control_cost():
    """Q_uu + mu 1 everywhere = the backward pass of the problem whose R is R + (mu / 2) 1, as long as
    C_u = 2 R (u - u_d) is the same, i.e. on a trajectory flown with the desired controls."""
    cfg = _hover_cfg(B=3, n=12)
    s = _oracle(cfg)
    mu = 0.75
    heavier = dict(cfg, R=cfg["R"] + 0.5 * mu * np.eye(4))
    h = _oracle(heavier)
    for b in range(3):
        traj = s.forward_sim(cfg["init"][b], np.zeros((12, 52)), 1.0)  # controls stay u_d, states move
        g_mu, t_mu = s.backwards_pass_reg(traj, mu)
        g_h, t_h = h.backwards_pass(traj)
        np.testing.assert_allclose(g_mu, g_h, rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(t_mu, t_h, rtol=1e-11)
        g0, t0 = s.backwards_pass_reg(traj, 0.0)
        g, t = s.backwards_pass(traj)
        assert np.array_equal(g0, g) and np.array_equal(t0, t)


def test_restarts_take_over_where_the_line_search_gives_up():
    """With one trial per line search (full steps only) most problems end in ilqr.hh:191-193; with restarts
    the same problems reach the optimum the back-tracking solver finds."""
    cfg = _hover_cfg()
    plain = _oracle(cfg).solve_batch(cfg["init"], n_threads=4)
    failed = plain["status"] == orc.STATUS_LINE_SEARCH_FAILED
    assert failed.sum() >= 8
    s = _oracle(cfg)
    s.set_regularisation(1.0, 4.0, 1e6)
    reg = s.solve_batch(cfg["init"], n_threads=4)
    assert (reg["status"] != orc.STATUS_LINE_SEARCH_FAILED).all()
    # problems that never exhausted a line search are untouched, bit for bit
    for k in ("traj", "cost", "iters", "n_bwd", "n_fwd", "status"):
        assert np.array_equal(reg[k][~failed], plain[k][~failed]), k
    # the others went on from the iterate they were stuck at: never worse, and restarts show in n_bwd
    assert (reg["cost"][failed] <= plain["cost"][failed]).all()
    assert (reg["n_bwd"][failed] > reg["iters"][failed] + 1).all()
    backtracking = _oracle(dict(cfg, options=dict(cfg["options"], ls_max_iters=100))).solve_batch(cfg["init"], n_threads=4)
    done = reg["status"] != orc.STATUS_MAX_ITERS
    np.testing.assert_allclose(reg["cost"][done], backtracking["cost"][done], rtol=1e-8)


def test_restarts_end_in_line_search_failure_past_mu_max():
    """No step satisfies a 10x Armijo demand whatever mu: mu_init, x factor ... until mu_max, then status 3;
    every restart costs one backward pass and ls_max_iters trials."""
    cfg = _hover_cfg(B=4, n=10, ls_max_iters=3)
    cfg["options"] = dict(cfg["options"], desired_reduction_frac=10.0)
    s = _oracle(cfg)
    s.set_regularisation(0.5, 2.0, 4.0)  # 0.5, 1, 2, 4: four restarts
    out = s.solve_batch(cfg["init"])
    assert (out["status"] == orc.STATUS_LINE_SEARCH_FAILED).all()
    np.testing.assert_array_equal(out["iters"], 1)
    np.testing.assert_array_equal(out["n_bwd"], 2 + 4)
    np.testing.assert_array_equal(out["n_fwd"], 1 + 3 * 5)
    with pytest.raises(ValueError):
        s.set_regularisation(1.0, 1.0, 10.0)  # factor must exceed 1
    with pytest.raises(ValueError):
        s.set_regularisation(1.0, 2.0, 0.5)   # mu_max below mu_init
    s.set_regularisation(0.0)                  # off again: the reference's behaviour
    off = s.solve_batch(cfg["init"])
    np.testing.assert_array_equal(off["n_bwd"], 2)
