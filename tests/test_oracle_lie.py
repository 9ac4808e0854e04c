"""Oracle SE(3)/SO(3) arithmetic vs an independent expm/logm implementation and FD.

manif (WORKSPACE:55-61) is not in the reference tree, so these checks pin the
oracle's restatement of SURVEY.md Appendix A to the group axioms themselves:
exp/log against scipy's matrix exponential/logarithm, every Jacobian against
central finite differences on the manifold.  Call sites that depend on them:
quadrotor_model.cc:183-186, 204, 211, 217, 232-235.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import liecheck as lc

RNG = np.random.default_rng(7)

TAUS = [
    np.array([1.0, 2.0, 3.0, 0.4, 0.5, 0.6]),
    np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0]) * 0.3,   # |theta| ~ 2.6
    np.array([0.3, -0.2, 0.1, 1e-3, -2e-3, 1.5e-3]),  # small but above the switch
    np.array([0.3, -0.2, 0.1, 1e-6, -2e-6, 1.5e-6]),  # below the theta^2 <= 1e-10 switch
    np.array([0.3, -0.2, 0.1, 0.0, 0.0, 0.0]),
    np.array([-1.0, 0.5, 2.0, 3.1, 0.0, 0.0]),        # close to pi about x
    np.array([-1.0, 0.5, 2.0, 0.0, 0.0, np.pi - 1e-4]),
] + [np.concatenate([RNG.uniform(-2, 2, 3), RNG.uniform(-1.7, 1.7, 3)]) for _ in range(8)]


@pytest.mark.parametrize("tau", TAUS)
def test_se3_exp_matches_expm(tau):
    T = orc.se3_exp(tau)
    M = lc.exp_mat(tau)
    # below the theta^2 <= 1e-10 switch manif truncates at first order: error <= theta^2/6 |rho|
    small = tau[3:] @ tau[3:] <= 1e-10
    atol = 2e-11 if small else 2e-14 * max(1.0, np.abs(M).max())
    np.testing.assert_allclose(lc.pose_to_mat(T), M, rtol=0, atol=atol)
    assert abs(np.linalg.norm(T[3:]) - 1.0) < 1e-10


@pytest.mark.parametrize("tau", TAUS)
def test_se3_log_inverts_exp(tau):
    T = orc.se3_exp(tau)
    back = orc.se3_log(T)
    small = tau[3:] @ tau[3:] <= 1e-10
    np.testing.assert_allclose(back, tau, rtol=0, atol=(4e-11 if small else 5e-12 * max(1.0, np.abs(tau).max())))
    np.testing.assert_allclose(back, lc.log_mat(lc.exp_mat(tau)), rtol=0, atol=1e-10)


def test_so3_log_negative_w_branch():
    # demo desired roll reaches pi (quadrotor_ilqr.py:101-106); -q is the same rotation
    th = np.array([2.5, 0.3, -0.4])
    q = orc.so3_exp(th)
    np.testing.assert_allclose(orc.so3_log(q), th, atol=1e-13)
    np.testing.assert_allclose(orc.so3_log(-q), th, atol=1e-13)
    q_pi = np.array([np.cos(np.pi / 2), np.sin(np.pi / 2), 0.0, 0.0])
    np.testing.assert_allclose(np.abs(orc.so3_log(q_pi)), [np.pi, 0, 0], atol=1e-13)


def test_so3_small_angle_branches():
    th = np.array([1e-6, -2e-6, 3e-6])
    q = orc.so3_exp(th)
    np.testing.assert_allclose(q, [1.0, 0.5e-6, -1e-6, 1.5e-6], atol=1e-20)  # [th/2, 1], not renormalised
    np.testing.assert_allclose(orc.so3_log(q), th, atol=1e-20)
    np.testing.assert_allclose(orc.so3_ljac(th), np.eye(3) + 0.5 * lc.hat3(th), atol=1e-20)
    np.testing.assert_allclose(orc.so3_ljacinv(th), np.eye(3) - 0.5 * lc.hat3(th), atol=1e-20)


@pytest.mark.parametrize("i", range(6))
def test_compose_inverse_adj_match_matrices(i):
    A = orc.se3_exp(TAUS[i])
    B = orc.se3_exp(TAUS[i + 7])
    MA, MB = lc.pose_to_mat(A), lc.pose_to_mat(B)
    np.testing.assert_allclose(lc.pose_to_mat(orc.se3_compose(A, B)), MA @ MB, atol=1e-13)
    np.testing.assert_allclose(lc.pose_to_mat(orc.se3_inverse(A)), np.linalg.inv(MA), atol=1e-13)
    # Ad(T) tau^ = T tau^ T^-1
    tau = TAUS[i + 2]
    lhs = lc.hat6(orc.se3_adj(A) @ tau)
    np.testing.assert_allclose(lhs, MA @ lc.hat6(tau) @ np.linalg.inv(MA), atol=1e-12)


def _fd_jac(fun, n_in, eps=1e-6):
    cols = []
    for i in range(n_in):
        d = np.zeros(n_in)
        d[i] = eps
        cols.append((fun(d) - fun(-d)) / (2 * eps))
    return np.stack(cols, axis=1)


@pytest.mark.parametrize("tau", TAUS[:3] + TAUS[7:11])
def test_se3_rjac_and_rjacinv_vs_fd(tau):
    # rjac: Exp(tau + d) = Exp(tau) Exp(Jr d)
    E = lc.exp_mat(tau)
    fd = _fd_jac(lambda d: lc.log_mat(np.linalg.inv(E) @ lc.exp_mat(tau + d)), 6)
    Jr = orc.se3_rjac(tau)
    np.testing.assert_allclose(Jr, fd, atol=2e-8)
    np.testing.assert_allclose(orc.se3_rjacinv(tau) @ Jr, np.eye(6), atol=1e-11)
    # ljacinv(tau) = rjacinv(-tau)
    np.testing.assert_allclose(orc.se3_ljacinv(tau), orc.se3_rjacinv(-tau), atol=0)
    # Jl = Ad(Exp tau) Jr
    Jl = orc.se3_adj(orc.se3_exp(tau)) @ Jr
    np.testing.assert_allclose(orc.se3_ljacinv(tau) @ Jl, np.eye(6), atol=1e-10)


def test_se3_jacobians_small_angle_continuity():
    rho = np.array([0.3, -0.2, 0.1])
    ax = np.array([1.0, -2.0, 1.5]) / np.linalg.norm([1.0, -2.0, 1.5])
    below = np.concatenate([rho, ax * 0.99e-5])  # theta^2 < 1e-10: series branch
    above = np.concatenate([rho, ax * 1.01e-5])  # closed-form branch
    for f in (orc.se3_rjac, orc.se3_rjacinv):
        np.testing.assert_allclose(f(below), f(above), atol=2e-6)  # cancellation-limited near the switch
        # and both agree with the first-order model I -/+ ad(tau)/2
        np.testing.assert_allclose(f(below)[:3, :3], f(below)[3:, 3:], atol=0)
