"""Where the device path deliberately leaves the reference's arithmetic, pinned by tests: the factorisation of Q_uu in the
symmetric-weight kernels (unpivoted LDL^T in registers against Eigen's diagonally pivoted ldlt(), ilqr.hh:126; the general
kernel -- non-symmetric weights, or force_general = 1 -- pivots as Eigen does) and fractional max_iters."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402  (the checker)
from quadrotorilqr_amd import capi, problems as pb  # noqa: E402


def _single_knot_problem(R, B=4, seed=3):
    """One knot: V = 0, so Q_uu = C_uu = 2 R exactly, Q_xu = 0, and k = -(2R)^-1 (2 R du) = -du in exact
    arithmetic whatever R is: the error of k measures the factorisation alone."""
    r = np.random.default_rng(seed)
    cfg = pb.config2(B=B, N=1, seed=seed)
    cfg = dict(cfg, R=R)
    init = cfg["init"].copy()
    init[:, 0, 14:18] += r.uniform(-1, 1, (B, 4))   # du != 0
    return cfg, init


def _k_error(gains, init, desired):
    k = gains[:, 0, :4]
    du = init[:, 0, 14:18] - desired[0, 14:18]
    return np.abs(k + du).max() / np.abs(du).max()


def test_spd_ill_conditioned_quu_matches_pivoted_ldlt():
    """SPD Q_uu with condition number 1e10: pivoted and unpivoted LDL^T are both backward stable for SPD matrices;
    the device gains agree with the oracle's (restated Eigen pivoted LDLT) to cond * eps."""
    r = np.random.default_rng(0)
    Qo, _ = np.linalg.qr(r.uniform(-1, 1, (4, 4)))
    R = Qo @ np.diag([1.0, 1e-3, 1e-6, 1e-10]) @ Qo.T
    R = (R + R.T) / 2
    cfg, init = _single_knot_problem(R)
    g, _ = capi.from_config(cfg).backwards_pass(init)
    ref = oracle_for(cfg)
    e_gpu = _k_error(g, init, cfg["desired"])
    e_ref = max(_k_error(ref.backwards_pass(init[b])[0][None], init[b:b + 1], cfg["desired"]) for b in range(len(init)))
    assert e_ref < 1e-4 and e_gpu < 1e-4          # cond(Q_uu) eps = 1e-6 at worst
    assert e_gpu < 100 * max(e_ref, 1e-9)


def oracle_for(cfg):
    return orc.OracleSolver(orc.model_params(**cfg["model"]), cfg["Q"], cfg["R"], cfg["desired"], cfg["dt"],
                            orc.options(**cfg["options"]))


def test_indefinite_quu_with_well_conditioned_minors_matches_pivoted_ldlt():
    """Indefinite R (one negative eigenvalue) whose leading principal minors are all well conditioned: LDL^T exists
    without pivoting, and in exact arithmetic equals the pivoted one; the device matches the oracle to rounding.
    (Neither the reference nor this library detects a Q_uu that is not positive definite: ilqr.hh:126-128 never
    looks at ldlt().info().)"""
    R = np.diag([1.0, 2.0, -0.5, 1.5])
    R[0, 1] = R[1, 0] = 0.3
    R[2, 3] = R[3, 2] = 0.2
    cfg, init = _single_knot_problem(R)
    g, terms = capi.from_config(cfg).backwards_pass(init)
    ref = oracle_for(cfg)
    for b in range(len(init)):
        g_ref, t_ref = ref.backwards_pass(init[b])
        np.testing.assert_allclose(g[b], g_ref, rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(terms[b], t_ref, rtol=1e-11)
    assert _k_error(g, init, cfg["desired"]) < 1e-13
    gg, tg = capi.from_config(cfg, force_general=1).backwards_pass(init)  # the pivoting kernel: other pivot order, same answer
    np.testing.assert_allclose(gg, g, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(tg, terms, rtol=1e-11)
    # and over a horizon (Q_uu = 2 R + J_u^T V_xx J_u stays indefinite near the end of the trajectory)
    cfg8 = dict(pb.config2(B=4, N=8, seed=3), R=R)
    g8, t8 = capi.from_config(cfg8).backwards_pass(cfg8["init"])
    for b in range(4):
        g_ref, t_ref = oracle_for(cfg8).backwards_pass(cfg8["init"][b])
        np.testing.assert_allclose(g8[b], g_ref, rtol=1e-8, atol=1e-9 * np.abs(g_ref).max())
        np.testing.assert_allclose(t8[b], t_ref, rtol=1e-8)


@pytest.mark.parametrize("pivot", [1e-4, 1e-8, 1e-12])
def test_tiny_leading_pivot_is_where_unpivoted_ldlt_leaves_eigen(pivot):
    """DOCUMENTED DEVIATION.  R = [[p, 1], [1, 1]] (+) I with p -> 0 is indefinite with a tiny LEADING entry.
    Eigen's ldlt() pivots on the largest remaining diagonal entry (it starts with R[1][1]) and stays accurate to
    rounding; the device's unpivoted LDL^T divides by p first and loses ~eps / p of relative accuracy.  The test pins
    both facts: the oracle's k = -du to 1e-13, the device's error bounded by 100 eps / p and -- for the smallest
    pivots -- visibly larger than the oracle's.  No symmetric positive definite R can produce this case (for SPD
    matrices the two factorisations are equally stable: first test)."""
    R = np.eye(4)
    R[0, 0] = pivot
    R[0, 1] = R[1, 0] = 1.0
    cfg, init = _single_knot_problem(R)
    g, _ = capi.from_config(cfg).backwards_pass(init)
    ref = oracle_for(cfg)
    e_gpu = _k_error(g, init, cfg["desired"])
    e_ref = max(_k_error(ref.backwards_pass(init[b])[0][None], init[b:b + 1], cfg["desired"]) for b in range(len(init)))
    assert e_ref < 1e-13
    assert np.isfinite(g).all() and e_gpu < 100 * np.finfo(float).eps / pivot
    if pivot <= 1e-8:
        assert e_gpu > 10 * e_ref   # this is where the two factorisations part ways
    # force_general = 1 selects the kernel that keeps the reference's forms, Eigen's diagonal pivoting included
    # (backward_layout.h, ldlt4_pivoted_solve): it starts with R[1][1] as Eigen does and stays at rounding level
    g1, t1 = capi.from_config(cfg, force_general=1).backwards_pass(init)
    e_gen = _k_error(g1, init, cfg["desired"])
    assert e_gen <= max(10 * e_ref, 1e-15)
    for b in range(len(init)):
        g_ref, t_ref = ref.backwards_pass(init[b])
        np.testing.assert_allclose(g1[b], g_ref, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(t1[b], t_ref, rtol=1e-11)


def test_fractional_max_iters_counts_like_the_reference_loop():
    """ConvergenceCriteria.max_iters is a double compared with an int counter (ilqr_options.hh:14, ilqr.hh:58):
    max_iters = 2.5 runs i = 0, 1, 2 -- three forward passes, three ILQRIterDebug entries."""
    cfg = pb.config2(B=5, N=30)
    cfg["options"] = dict(cfg["options"], max_iters=2.5, populate_debug=True)
    s = capi.from_config(cfg)
    out = s.solve_batch(cfg["init"])
    ref = oracle_for(cfg).solve_batch(cfg["init"])
    np.testing.assert_array_equal(out["iters"], 3)
    np.testing.assert_array_equal(out["iters"], ref["iters"])
    np.testing.assert_array_equal(out["status"], ref["status"])
    hist = s.cost_history(5)
    assert hist.shape[1] == 3 and np.isfinite(hist).all()
    np.testing.assert_array_equal(hist[:, 2], out["cost"])
    traj, info = s.solve(cfg["init"][0])
    assert info["iters"] == 3 and len(info["debug_costs"]) == 3 and len(info["debug_trajs"]) == 3
    np.testing.assert_array_equal(info["debug_trajs"][-1], traj)


def test_a_lost_hand_off_in_the_rollout_ends_the_call_with_an_error_not_a_hang():
    """k_rollout16's wavefronts hand values to each other through LDS progress words; every wait is a BOUNDED spin, so that
    a hand-off that never comes drains the grid instead of hanging the GPU.  The diagnostics build can withhold one
    (qilqr_debug_set_rollout_stall): the waiting wavefront's spin runs out, the block raises its abort word, every role
    leaves, the kernel reports the block through pinned memory, and the host turns that into QILQR_ERR_HIP -- the call
    fails loudly, nothing iterates on the stale candidate, and the handle works again once the fault is removed."""
    import time
    from tests.diag_lib import capi_diag
    d = capi_diag()
    cfg = pb.config2(B=8, N=30, seed=5)
    s = d.from_config(cfg, single_wave_rollout=3)           # k_rollout16 at every size
    good = s.solve_batch(cfg["init"])
    assert np.isin(good["status"], [0, 1]).all()
    gains = np.zeros((8, 30, 52))
    rolled = s.forward_sim(cfg["init"], gains, 1.0)
    lib = d.load()
    assert lib.qilqr_debug_set_rollout_stall(s._h, 7) == 0    # the velocity hand-off of knot 7 is never announced
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="hand-off between the wavefronts of block"):
        s.forward_sim(cfg["init"], gains, 1.0)
    with pytest.raises(RuntimeError, match="results of this call are invalid"):
        s.solve_batch(cfg["init"])
    assert time.perf_counter() - t0 < 60.0                     # bounded: a fraction of a second per abandoned launch
    assert lib.qilqr_debug_set_rollout_stall(s._h, -1) == 0
    np.testing.assert_array_equal(s.forward_sim(cfg["init"], gains, 1.0), rolled)
    again = s.solve_batch(cfg["init"])
    for k in ("status", "iters", "cost", "traj"):
        np.testing.assert_array_equal(again[k], good[k], err_msg=k)


def test_a_withheld_record_in_the_backward_pass_ends_the_call_with_an_error_not_a_hang():
    """k_backward4's fused form has no block barrier in its knot loop: the loader wavefront tags every ring slot it fills and
    the four matrix wavefronts read the tag in front of the operands; both sides wait in BOUNDED spins.  The diagnostics build
    can withhold one record's tags (qilqr_debug_set_backward_stall): the matrix wavefronts' waits run out, they finish the
    loop unchecked so that the grid drains, the block reports itself through pinned memory and the host returns
    QILQR_ERR_HIP; with the fault removed the handle gives the results it gave before."""
    import time
    from tests.diag_lib import capi_diag
    d = capi_diag()
    cfg = pb.config2(B=8, N=30, seed=5)
    s = d.from_config(cfg, force_general=5)
    good = s.solve_batch(cfg["init"])
    assert np.isin(good["status"], [0, 1]).all()
    trajs = s.forward_sim(cfg["init"], np.zeros((8, 30, 52)), 1.0)
    gains = s.backwards_pass(trajs)
    lib = d.load()
    assert lib.qilqr_debug_set_backward_stall(s._h, 11) == 0   # the tags of the twelfth record from the end never appear
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="k_backward4: a hand-off between the wavefronts of block"):
        s.backwards_pass(trajs)
    with pytest.raises(RuntimeError, match="results of this call are invalid"):
        s.solve_batch(cfg["init"])
    assert time.perf_counter() - t0 < 60.0
    assert lib.qilqr_debug_set_backward_stall(s._h, -1) == 0
    again_gains = s.backwards_pass(trajs)
    for a, b in zip(again_gains if isinstance(again_gains, tuple) else (again_gains,), gains if isinstance(gains, tuple) else (gains,)):
        np.testing.assert_array_equal(a, b)
    again = s.solve_batch(cfg["init"])
    for k in ("status", "iters", "cost", "traj"):
        np.testing.assert_array_equal(again[k], good[k], err_msg=k)


def test_the_product_build_refuses_the_kernels_of_the_diagnostics_build():
    cfg = pb.config2(B=4, N=10)
    with pytest.raises(TypeError, match="diagnostics build"):
        capi.from_config(cfg, persistent=1)
    with pytest.raises(TypeError, match="diagnostics build"):
        capi.from_config(cfg, force_general=3)
