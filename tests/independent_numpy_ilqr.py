"""An INDEPENDENT second restatement of the reference algorithm, for cross-checking the oracle.

Test infrastructure (like oracle/): nothing in the product imports it.  It follows the same reference lines
as oracle/ilqr_oracle.c (src/ilqr.hh:53-205, src/cost.hh:36-61, src/quadrotor_model.cc:33-122, 174-250,
266-276) but shares NO code and NO closed-form Lie-group formula with oracle/ or with
quadrotorilqr_amd/csrc/se3_math.h, so that an error common to those two (written by one author from one
reading of manif) cannot hide here:

  * poses are 4x4 homogeneous matrices, not (t, quaternion) pairs;
  * SE(3) Exp is scipy.linalg.expm of the 4x4 twist matrix (Pade approximation, no Rodrigues formula);
  * the rotation part of Log is scipy.spatial.transform.Rotation.as_rotvec; its translation part solves with
    V(theta) = phi1(hat(theta)), phi1(A) = int_0^1 exp(sA) ds, read off expm([[A, I],[0, 0]]);
  * the SE(3) Jacobians are J_l(tau) = phi1(ad(tau)) from the same block-matrix exponential on the 6x6 adjoint
    representation (no Barfoot Q block, no series), J_r(tau) = J_l(-tau), inverses by numpy.linalg.inv;
  * Q_uu systems are solved by numpy.linalg.solve (LU with partial pivoting), not LDL^T.

Tangent order is manif's [rho ; theta]; knots are the 18-column rows of quadrotorilqr_amd/problems.py.
"""
import numpy as np
from scipy.linalg import expm
from scipy.spatial.transform import Rotation


def hat3(a):
    return np.array([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]])


def twist_matrix(tau):
    M = np.zeros((4, 4))
    M[:3, :3] = hat3(tau[3:])
    M[:3, 3] = tau[:3]
    return M


def phi1(A):
    """int_0^1 exp(s A) ds for a square A, as a block of one matrix exponential"""
    n = A.shape[0]
    M = np.zeros((2 * n, 2 * n))
    M[:n, :n] = A
    M[:n, n:] = np.eye(n)
    return expm(M)[:n, n:]


def ad(tau):
    """adjoint representation of the twist [rho ; theta] on se(3) in that order"""
    W, V = hat3(tau[3:]), hat3(tau[:3])
    return np.block([[W, V], [np.zeros((3, 3)), W]])


def Ad(T):
    R, t = T[:3, :3], T[:3, 3]
    return np.block([[R, hat3(t) @ R], [np.zeros((3, 3)), R]])


def se3_exp(tau):
    return expm(twist_matrix(tau))


def se3_log(T):
    th = Rotation.from_matrix(T[:3, :3]).as_rotvec()
    rho = np.linalg.solve(phi1(hat3(th)), T[:3, 3])
    return np.concatenate([rho, th])


def se3_left_jacobian(tau):
    return phi1(ad(tau))


def se3_right_jacobian(tau):
    return phi1(ad(-np.asarray(tau)))


def pose_from_knot(p):
    T = np.eye(4)
    T[:3, :3] = Rotation.from_quat([p[5], p[6], p[7], p[4]]).as_matrix()
    T[:3, 3] = p[1:4]
    return T


def knot_from_state(time_s, T, v, u):
    q = Rotation.from_matrix(T[:3, :3]).as_quat()  # x, y, z, w
    return np.concatenate([[time_s], T[:3, 3], [q[3], q[0], q[1], q[2]], v, u])


class Model:
    """QuadrotorModel (quadrotor_model.cc:6-25)"""

    def __init__(self, mass_kg, inertia, arm_length_m, torque_to_thrust_ratio_m, g_mpss):
        self.m, self.I, self.g = mass_kg, np.asarray(inertia, dtype=float), g_mpss
        a, c = arm_length_m, torque_to_thrust_ratio_m
        self.arms = np.array([[0, -a, 0, a], [a, 0, -a, 0], [-c, c, -c, c]], dtype=float)

    def continuous(self, T, v, u, diffs=False):
        """quadrotor_model.cc:65-122: body acceleration, and its Jacobians"""
        R = T[:3, :3]
        ez = np.array([0.0, 0.0, 1.0])
        w = v[3:]
        lin = -self.g * (R.T @ ez) + u.sum() * ez / self.m  # no -w x v term: as the reference
        ang = np.linalg.solve(self.I, self.arms @ u - hat3(w) @ self.I @ w)
        acc = np.concatenate([lin, ang])
        if not diffs:
            return acc
        Jx = np.zeros((12, 12))
        Jx[0:6, 6:12] = np.eye(6)
        Jx[6:9, 3:6] = -self.g * hat3(R.T @ ez)
        Jx[9:12, 9:12] = -np.linalg.solve(self.I, hat3(w) @ self.I - hat3(self.I @ w))
        Ju = np.zeros((12, 4))
        Ju[8, :] = 1.0 / self.m
        Ju[9:12, :] = np.linalg.solve(self.I, self.arms)
        return acc, Jx, Ju

    def step(self, T, v, u, dt, diffs=False):
        """discrete_dynamics (quadrotor_model.cc:33-49) = euler_step (266-276) of the continuous dynamics"""
        if not diffs:
            acc = self.continuous(T, v, u)
            return T @ se3_exp(dt * v), v + dt * acc
        acc, Jcx, Jcu = self.continuous(T, v, u, True)
        tau = dt * v
        E = se3_exp(tau)
        J_lhs = np.eye(12)
        J_lhs[:6, :6] = np.linalg.inv(Ad(E))          # d (T (+) tau) / dT          (quadrotor_model.cc:188-191)
        J_rhs = np.eye(12)
        J_rhs[:6, :6] = se3_right_jacobian(tau)        # d (T (+) tau) / dtau        (:193-195)
        J_rhs = dt * J_rhs                             # euler_step scales the whole block (:272)
        return (T @ E, v + dt * acc), J_lhs + J_rhs @ Jcx, J_rhs @ Jcu

    # ---- the Runge-Kutta extension: the step sketched in the comment at quadrotor_model.cc:51-63, from THIS file's primitives
    def _euler(self, T, v, k, h, diffs=False):
        """detail::euler_step (quadrotor_model.cc:266-276) of the state (T, v) along the tangent k = [pose rate ; acceleration]"""
        tau = h * k[:6]
        E = se3_exp(tau)
        out = (T @ E, v + h * k[6:])
        if not diffs:
            return out
        J_lhs = np.eye(12)
        J_lhs[:6, :6] = np.linalg.inv(Ad(E))
        J_rhs = np.eye(12)
        J_rhs[:6, :6] = se3_right_jacobian(tau)
        return out, J_lhs, h * J_rhs

    def step_rk4(self, T, v, u, dt, diffs=False):
        """k = 0, x_dot = 0; for (c, h) in ((1/6, 0), (2/6, dt/2), (2/6, dt/2), (1/6, dt)): k = f(euler_step(x, k, h), u);
        x_dot += c k; then x (+) dt x_dot.  Jacobians: the chain rule through the stages."""
        k, xdot = np.zeros(12), np.zeros(12)
        K, Ku, SK, SKu = np.zeros((12, 12)), np.zeros((12, 4)), np.zeros((12, 12)), np.zeros((12, 4))
        for c, h in ((1 / 6, 0.0), (2 / 6, dt / 2), (2 / 6, dt / 2), (1 / 6, dt)):
            if diffs:
                (Ti, vi), El, Er = self._euler(T, v, k, h, True)
                A, Bm = El + Er @ K, Er @ Ku
                acc, Fx, Fu = self.continuous(Ti, vi, u, True)
                K, Ku = Fx @ A, Fx @ Bm + Fu
                SK, SKu = SK + c * K, SKu + c * Ku
            else:
                Ti, vi = self._euler(T, v, k, h)
                acc = self.continuous(Ti, vi, u)
            k = np.concatenate([vi, acc])
            xdot = xdot + c * k
        if not diffs:
            return self._euler(T, v, xdot, dt)
        out, El, Er = self._euler(T, v, xdot, dt, True)
        return out, El + Er @ SK, Er @ SKu


def cost_knot(Q, R, T, v, u, Td, vd, ud, diffs=False):
    """CostFunction::operator() (cost.hh:36-61)"""
    tau = se3_log(np.linalg.inv(Td) @ T)               # x (-) x_d, quadrotor_model.cc:221-235
    dx = np.concatenate([tau, v - vd])
    du = u - ud
    c = dx @ Q @ dx + du @ R @ du
    if not diffs:
        return c
    J = np.eye(12)
    J[:6, :6] = np.linalg.inv(se3_right_jacobian(tau))  # d (x (-) x_d) / dx = J_r^-1(tau)
    return c, dict(x=2 * (dx @ Q @ J), u=2 * (du @ R), xx=2 * J.T @ Q @ J, uu=2 * R, xu=np.zeros((12, 4)))


class ILQR:
    """ILQR<QuadrotorModel> (ilqr.hh:25-206) on (n, 18) knot arrays"""

    def __init__(self, model, Q, R, desired, dt, options, integrator=0, recursion=0):
        # recursion 1: the substituted, symmetrised value update (the extension orc_set_recursion of the oracle states)
        self.recursion = recursion
        self.model, self.Q, self.R, self.dt, self.o = model, np.asarray(Q, float), np.asarray(R, float), dt, options
        self.step = model.step_rk4 if integrator == 1 else model.step  # 1: the Runge-Kutta extension
        self.des = [(pose_from_knot(p), p[8:14].copy(), p[14:18].copy()) for p in np.asarray(desired)]

    @staticmethod
    def unpack(traj):
        return [(pose_from_knot(p), p[8:14].copy(), p[14:18].copy()) for p in traj]

    def cost_trajectory(self, pts):  # ilqr.hh:89-95
        c = 0.0
        for i, (T, v, u) in enumerate(pts):
            c += cost_knot(self.Q, self.R, T, v, u, *self.des[i])
        return c

    def backwards_pass(self, pts):  # ilqr.hh:97-147
        n = len(pts)
        vx, vxx = np.zeros(12), np.zeros((12, 12))
        ks, Ks = [None] * n, [None] * n
        QuTk = kTQuuk = 0.0
        for i in range(n - 1, -1, -1):
            T, v, u = pts[i]
            _, Jx, Ju = self.step(T, v, u, self.dt, True)
            _, C = cost_knot(self.Q, self.R, T, v, u, *self.des[i], diffs=True)
            Qx = C["x"] + Jx.T @ vx
            Qu = C["u"] + Ju.T @ vx
            Qxx = C["xx"] + Jx.T @ vxx @ Jx
            Quu = C["uu"] + Ju.T @ vxx @ Ju
            Qxu = C["xu"] + Jx.T @ vxx @ Ju
            K = -np.linalg.solve(Quu, Qxu.T)
            k = -np.linalg.solve(Quu, Qu)
            ks[i], Ks[i] = k, K
            QuTk += Qu @ k
            if self.recursion == 1:           # EXTENSION: K = -Quu^-1 Qux, k = -Quu^-1 Qu substituted, V_xx symmetrised
                vx = Qx + K.T @ Qu
                vxx = Qxx + Qxu @ K
                vxx = 0.5 * (vxx + vxx.T)
                kTQuuk += -(Qu @ k)
                continue
            vx = Qx - K.T @ Quu @ k
            vxx = Qxx - K.T @ Quu @ K       # not symmetrised, as the reference
            kTQuuk += k @ Quu @ k
        return ks, Ks, (QuTk, kTQuuk)

    def forward_sim(self, pts, ks, Ks, alpha):  # ilqr.hh:149-172
        out = []
        T, v = pts[0][0].copy(), pts[0][1].copy()
        for i, (Tn, vn, un) in enumerate(pts):
            dx = np.concatenate([se3_log(np.linalg.inv(Tn) @ T), v - vn])
            u = un + alpha * ks[i] + Ks[i] @ dx
            out.append((T, v, u))
            T, v = self.step(T, v, u, self.dt)
        return out

    def is_converged(self, cost, new):  # ilqr.hh:196-205
        with np.errstate(divide="ignore", invalid="ignore"):
            if abs(cost - new) / abs(cost) < self.o["rtol"]:
                return True
        return abs(cost - new) < self.o["atol"]

    def solve(self, traj):  # ilqr.hh:53-87
        traj = np.asarray(traj, dtype=float)
        times = traj[:, 0].copy()
        pts = self.unpack(traj)
        new_cost = self.cost_trajectory(pts)
        hist, n_bwd, n_fwd, status, i = [], 0, 0, 2, 0
        while i < self.o["max_iters"]:
            ks, Ks, (a, b) = self.backwards_pass(pts)
            n_bwd += 1
            cost = new_cost
            if i > 0 and self.is_converged(cost, cost + a + b / 2.0):
                status = 0
                break
            if i == 0:
                pts = self.forward_sim(pts, ks, Ks, 1.0)
                new_cost = self.cost_trajectory(pts)
                n_fwd += 1
            else:
                step, found = 1.0, False
                for _ in range(self.o["ls_max_iters"]):
                    cand = self.forward_sim(pts, ks, Ks, step)
                    c = self.cost_trajectory(cand)
                    n_fwd += 1
                    if c - cost < self.o["desired_reduction_frac"] * (step * a + step * step * b / 2.0):
                        pts, new_cost, found = cand, c, True
                        break
                    step *= self.o["step_update"]
                if not found:
                    status = 3
                    break
            hist.append(new_cost)
            i += 1
            if i - 1 > 0 and self.is_converged(cost, new_cost):
                status = 1
                break
        out = np.array([knot_from_state(times[j], T, v, u) for j, (T, v, u) in enumerate(pts)])
        return dict(traj=out, cost=new_cost, status=status, iters=len(hist), n_bwd=n_bwd, n_fwd=n_fwd,
                    cost_hist=np.array(hist))
