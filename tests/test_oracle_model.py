"""Oracle vs the reference's quadrotor_model_test.cc and cost_test.cc cases.

Each test names the reference test it restates (file:line).  Inputs are the
reference's own (x = Exp([1..6]), v = [2..7], tangents [3..8]/[4..9], rhs =
Exp(2*[1..6])); the only substitution is the random SPD inertia, which the
reference draws from srand(0)+Eigen::Random (not reproducible offline) and we
draw from a seeded NumPy generator with the same A A^T + 3 I construction.
"""
import numpy as np
import pytest

from oracle import oracle as orc

DT = 0.1
MASS = 1.0


def random_inertia():
    A = np.random.default_rng(0).uniform(-1, 1, (3, 3))  # quadrotor_model_test.cc:22-28
    return A @ A.T + 3 * np.eye(3)


def quad(inertia=None, ttr=1.0, g=9.81):
    return orc.model_params(MASS, np.eye(3) if inertia is None else inertia, 1.0, ttr, g)


def state(tau, v):
    return np.concatenate([orc.se3_exp(np.asarray(tau, float)), np.asarray(v, float)])


X_INIT = state([1, 2, 3, 4, 5, 6], [2, 3, 4, 5, 6, 7])
IDENT = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0.0])
EPS = 1e-6


def check_state_jacobian(fun, analytic):
    """quadrotor_model_test.cc:31-54: per column rel < 1e-2 or abs < 1e-12, step 1e-6.
    fun(delta[12]) returns a *state*; differences are taken with the model's (-)."""
    for i in range(12):
        d = np.zeros(12)
        d[i] = EPS
        fd = orc.state_minus(fun(d), fun(-d)) / (2 * EPS)
        col = analytic[:, i]
        err = np.linalg.norm(col - fd)
        assert err / max(np.linalg.norm(col), 1e-300) < 0.01 or err < 1e-12, (i, col, fd)
        # the build's own, tighter bar
        assert err < 1e-6 * max(1.0, np.linalg.norm(col)), (i, err)


def check_tangent_jacobian(fun, analytic, n_in):
    """same, for functions returning a 12-vector"""
    for i in range(n_in):
        d = np.zeros(n_in)
        d[i] = EPS
        fd = (fun(d) - fun(-d)) / (2 * EPS)
        col = analytic[:, i]
        err = np.linalg.norm(col - fd)
        assert err / max(np.linalg.norm(col), 1e-300) < 0.01 or err < 1e-12, (i, col, fd)
        assert err < 1e-6 * max(1.0, np.linalg.norm(col)), (i, err)


# ---- quadrotor_model_test.cc:94-116
def test_discrete_dynamics_updates_translational_states():
    x = IDENT.copy()
    x[7:10] = [1.0, 2.0, 3.0]
    u = np.ones(4)
    xn = orc.discrete_dynamics(quad(), x, u, DT)
    np.testing.assert_allclose(xn[:3], [0.1, 0.2, 0.3], rtol=1e-6)        # pose uses the OLD velocity
    np.testing.assert_allclose(xn[7:10], [1.0, 2.0, 3.0 + (4.0 / MASS - 9.81) * DT], rtol=1e-6)
    np.testing.assert_allclose(xn[10:13], 0.0, atol=1e-15)
    np.testing.assert_allclose(xn[3:7], [1, 0, 0, 0], atol=1e-15)


# ---- quadrotor_model_test.cc:118-143
def test_discrete_dynamics_updates_rotational_states():
    x = IDENT.copy()
    x[10:13] = [1.2, 0.0, 0.0]
    u = np.array([0.0, -1.0, 0.0, 1.0])
    xn = orc.discrete_dynamics(quad(), x, u, DT)
    expected_q = orc.so3_exp([0.12, 0, 0])
    assert np.linalg.norm(orc.so3_log(xn[3:7]) - orc.so3_log(expected_q)) < 1e-6
    np.testing.assert_allclose(xn[10:13], [1.2 + 2.0 * DT, 0, 0], rtol=1e-6, atol=1e-15)


# ---- quadrotor_model_test.cc:145-171, 173-199
def test_discrete_dynamics_state_jacobian_vs_fd():
    q = quad(random_inertia())
    u = np.zeros(4)
    _, Jx, _ = orc.discrete_dynamics(q, X_INIT, u, DT, diffs=True)
    check_state_jacobian(lambda d: orc.discrete_dynamics(q, orc.state_add(X_INIT, d), u, DT), Jx)


def test_discrete_dynamics_control_jacobian_vs_fd():
    q = quad(random_inertia())
    u = np.array([1.0, 2.0, 3.0, 4.0])
    _, _, Ju = orc.discrete_dynamics(q, X_INIT, u, DT, diffs=True)
    for i in range(4):
        d = np.zeros(4)
        d[i] = EPS
        fd = orc.state_minus(orc.discrete_dynamics(q, X_INIT, u + d, DT),
                             orc.discrete_dynamics(q, X_INIT, u - d, DT)) / (2 * EPS)
        err = np.linalg.norm(Ju[:, i] - fd)
        assert err / np.linalg.norm(Ju[:, i]) < 0.01 or err < 1e-12
        assert err < 1e-7


# ---- quadrotor_model_test.cc:201-224, 226-249
def test_continuous_dynamics_jacobians_vs_fd():
    q = quad(random_inertia())
    u0 = np.zeros(4)
    _, Jx, _ = orc.continuous_dynamics(q, X_INIT, u0, diffs=True)
    check_tangent_jacobian(lambda d: orc.continuous_dynamics(q, orc.state_add(X_INIT, d), u0), Jx, 12)
    u = np.array([1.0, 2.0, 3.0, 4.0])
    _, _, Ju = orc.continuous_dynamics(q, X_INIT, u, diffs=True)
    check_tangent_jacobian(lambda d: orc.continuous_dynamics(q, X_INIT, u + d), Ju, 4)


def test_control_jacobian_is_the_constant_the_design_relies_on():
    # J_u rows 0-7 zero, row 8 = dt/m, rows 9-11 = dt I^-1 moment_arms
    I = random_inertia()
    q = quad(I, ttr=0.3)
    _, _, Ju = orc.discrete_dynamics(q, X_INIT, np.array([1.0, 2, 3, 4]), DT, diffs=True)
    arms = np.array([[0, -1, 0, 1], [1, 0, -1, 0], [-0.3, 0.3, -0.3, 0.3]])
    expect = np.zeros((12, 4))
    expect[8] = DT / MASS
    expect[9:] = DT * np.linalg.solve(I, arms)
    np.testing.assert_allclose(Ju, expect, atol=1e-15)


# ---- quadrotor_model_test.cc:251-296
TANGENT = np.array([3, 4, 5, 6, 7, 8, 4, 5, 6, 7, 8, 9.0])


def test_state_add_jacobians_vs_fd():
    _, Jl, Jr = orc.state_add(X_INIT, TANGENT, diffs=True)
    check_state_jacobian(lambda d: orc.state_add(orc.state_add(X_INIT, d), TANGENT), Jl)
    check_state_jacobian(lambda d: orc.state_add(X_INIT, TANGENT + d), Jr)


# ---- quadrotor_model_test.cc:298-346
def test_state_minus_jacobians_vs_fd():
    lhs = X_INIT
    rhs = state(2 * np.arange(1, 7.0), 2 * np.arange(2, 8.0))
    _, Jl, Jr = orc.state_minus(lhs, rhs, diffs=True)
    check_tangent_jacobian(lambda d: orc.state_minus(orc.state_add(lhs, d), rhs), Jl, 12)
    check_tangent_jacobian(lambda d: orc.state_minus(lhs, orc.state_add(rhs, d)), Jr, 12)


# ---- quadrotor_model_test.cc:399-447
def test_euler_step_jacobians_vs_fd():
    _, Jl, Jr = orc.euler_step(X_INIT, TANGENT, DT, diffs=True)
    check_state_jacobian(lambda d: orc.euler_step(orc.state_add(X_INIT, d), TANGENT, DT), Jl)
    check_state_jacobian(lambda d: orc.euler_step(X_INIT, TANGENT + d, DT), Jr)


# ---- quadrotor_model.cc:19-24
def test_bad_inertia_is_rejected():
    assert orc.lib().orc_model_check(orc.C.byref(quad(np.diag([1.0, -1.0, 1.0])))) == orc.ERR_BAD_INERTIA
    asym = np.eye(3)
    asym[0, 1] = 0.1
    assert orc.lib().orc_model_check(orc.C.byref(quad(asym))) == orc.ERR_BAD_INERTIA
    assert orc.lib().orc_model_check(orc.C.byref(quad(random_inertia()))) == 0


# ======================= cost_test.cc =======================
def random_point(seed):
    r = np.random.default_rng(seed)
    x = state(np.concatenate([r.uniform(-1, 1, 3), r.uniform(-1, 1, 3)]), r.uniform(-1, 1, 6))
    return x, r.uniform(-1, 1, 4)


# ---- cost_test.cc:27-39
def test_zero_cost_when_zero_error():
    x, u = random_point(1)
    assert orc.cost(np.eye(12), np.eye(4), x, u, x, u) == 0.0


@pytest.mark.parametrize("seed", [2, 3])
def test_cost_differentials_vs_fd(seed):
    # cost_test.cc:66-150: Q = I12, R = I4, desired = a random point, evaluated AT the desired
    # point in the reference; here additionally away from it, and with a dense non-symmetric Q
    xd, ud = random_point(seed)
    x, u = random_point(seed + 10)
    r = np.random.default_rng(seed)
    for Q, R, (xx, uu) in [
        (np.eye(12), np.eye(4), (xd, ud)),
        (np.eye(12), np.eye(4), (x, u)),
        (r.uniform(-1, 1, (12, 12)) + 6 * np.eye(12), r.uniform(-1, 1, (4, 4)) + 3 * np.eye(4), (x, u)),
    ]:
        c, D = orc.cost(Q, R, xx, uu, xd, ud, diffs=True)
        assert c == orc.cost(Q, R, xx, uu, xd, ud)
        Z12, Z4 = np.zeros(12), np.zeros(4)
        f = lambda dx, du: orc.cost(Q, R, orc.state_add(xx, dx), uu + du, xd, ud)
        Cx_fd = np.zeros(12)
        for i in range(12):
            d = np.zeros(12)
            d[i] = EPS
            Cx_fd[i] = (f(d, Z4) - f(-d, Z4)) / (2 * EPS)
        Cu_fd = np.zeros(4)
        for i in range(4):
            d = np.zeros(4)
            d[i] = EPS
            Cu_fd[i] = (f(Z12, d) - f(Z12, -d)) / (2 * EPS)
        symQ = np.allclose(Q, Q.T)
        if symQ:  # for non-symmetric Q the reference's C_x = 2 dx^T Q J is NOT the gradient; kept as is
            assert np.linalg.norm(D["x"] - Cx_fd) / 12 < 1e-6  # cost_test.cc:79-80
            assert np.linalg.norm(D["u"] - Cu_fd) / 4 < 1e-6   # cost_test.cc:122-123
        np.testing.assert_array_equal(D["uu"], 2 * R)          # cost.hh:55
        np.testing.assert_array_equal(D["xu"], 0.0)            # cost.hh:57
        # cost.hh:51-52 restated with numpy from the oracle's own (-) Jacobian
        dx, J, _ = orc.state_minus(xx, xd, diffs=True)
        np.testing.assert_allclose(D["x"], 2 * dx @ Q @ J, rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(D["xx"], 2 * J.T @ Q @ J, rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(D["u"], 2 * (uu - ud) @ R, rtol=1e-13, atol=1e-13)


def test_gauss_newton_hessian_is_exact_at_zero_error():
    # cost_test.cc:83-107 evaluates C_xx at zero error, where Gauss-Newton is exact
    xd, ud = random_point(5)
    Q = np.eye(12)
    c, D = orc.cost(Q, np.eye(4), xd, ud, xd, ud, diffs=True)
    h = 1e-4
    H = np.zeros((12, 12))
    f = lambda d: orc.cost(Q, np.eye(4), orc.state_add(xd, d), ud, xd, ud)
    for i in range(12):
        for j in range(12):
            di, dj = np.zeros(12), np.zeros(12)
            di[i], dj[j] = h, h
            H[i, j] = (f(di + dj) - f(di - dj) - f(-di + dj) + f(-di - dj)) / (4 * h * h)
    np.testing.assert_allclose(D["xx"], H, atol=1e-5)
    assert np.linalg.norm(np.linalg.inv(D["xx"]) @ H - np.eye(12)) < 11.0  # the reference's own bar
