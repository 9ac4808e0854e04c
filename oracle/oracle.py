"""ctypes front-end of the CPU oracle (oracle/ilqr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() as the checker.  The product path never imports this.

Layouts (see ilqr_oracle.h): state x[13] = [t, q(w,x,y,z), v(6)]; knot p[18] =
[time, x, u]; gains g[52] = [k(4), K(4x12) column-major]; matrices row-major.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libilqr_oracle.so")

STATUS_CONVERGED_EXPECTED = 0
STATUS_CONVERGED = 1
STATUS_MAX_ITERS = 2
STATUS_LINE_SEARCH_FAILED = 3

ERR_BAD_INERTIA = 1
ERR_LENGTH_MISMATCH = 2
ERR_INVALID = 3


class ModelParams(C.Structure):
    _fields_ = [
        ("mass_kg", C.c_double),
        ("inertia", C.c_double * 9),
        ("arm_length_m", C.c_double),
        ("torque_to_thrust_ratio_m", C.c_double),
        ("g_mpss", C.c_double),
    ]


class Decision(C.Structure):
    """orc_decision (ilqr_oracle.h): one comparison a solve's control flow depended on, with its margin"""
    _fields_ = [("kind", C.c_int), ("iter", C.c_int), ("trial", C.c_int), ("result", C.c_int), ("n_bwd", C.c_int),
                ("n_fwd", C.c_int), ("lhs", C.c_double), ("rhs", C.c_double), ("margin", C.c_double)]


DEC_EXPECTED, DEC_ARMIJO, DEC_CONVERGED = 0, 1, 2


class Options(C.Structure):
    _fields_ = [
        ("step_update", C.c_double),
        ("desired_reduction_frac", C.c_double),
        ("ls_max_iters", C.c_int),
        ("rtol", C.c_double),
        ("atol", C.c_double),
        ("max_iters", C.c_double),
        ("populate_debug", C.c_int),
    ]


def build(force=False):
    """Compile the oracle with its Makefile (building the checker is not using it)."""
    src = os.path.join(_HERE, "ilqr_oracle.c")
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    ):
        subprocess.check_call(["make", "-C", _HERE, "libilqr_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def fast_library(native=True):
    """The TIMING-ONLY build of the same source (fused multiply-adds allowed; oracle/Makefile): bench.py's cpu_baseline.  Tries to
    compile it for this host's own CPU first (`make fast-native`), falls back to the portable build that travels with the tree.
    Never a parity comparand: pass it to OracleSolver(library=...) explicitly."""
    paths = []
    if native:
        try:
            subprocess.check_call(["make", "-C", _HERE, "fast-native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            paths.append(os.path.join(_HERE, "libilqr_oracle_fast_native.so"))
        except Exception:
            pass
    paths.append(os.path.join(_HERE, "libilqr_oracle_fast.so"))
    for p in paths:
        if not os.path.exists(p) and p.endswith("_fast.so"):
            subprocess.check_call(["make", "-C", _HERE, "libilqr_oracle_fast.so"], stdout=subprocess.DEVNULL)
        if os.path.exists(p):
            L = C.CDLL(p)
            L.orc_cost.restype = C.c_double
            L.orc_build_flavour.restype = C.c_char_p
            return L
    raise OSError("no fast oracle library")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_cost.restype = C.c_double
        _lib.orc_build_flavour.restype = C.c_char_p
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int)) if a is not None else None


def model_params(mass_kg, inertia, arm_length_m, torque_to_thrust_ratio_m, g_mpss=9.81):
    mp = ModelParams()
    mp.mass_kg = mass_kg
    I = _d(inertia).reshape(9)
    for i in range(9):
        mp.inertia[i] = I[i]
    mp.arm_length_m = arm_length_m
    mp.torque_to_thrust_ratio_m = torque_to_thrust_ratio_m
    mp.g_mpss = g_mpss
    return mp


def options(step_update=0.5, desired_reduction_frac=0.5, ls_max_iters=100, rtol=1e-12,
            atol=1e-12, max_iters=100, populate_debug=False):
    o = Options()
    o.step_update = step_update
    o.desired_reduction_frac = desired_reduction_frac
    o.ls_max_iters = int(ls_max_iters)
    o.rtol = rtol
    o.atol = atol
    o.max_iters = float(max_iters)
    o.populate_debug = int(bool(populate_debug))
    return o


# ---------------------------------------------------------------- Lie group
def _call_vec(name, n_out, *ins):
    out = np.zeros(n_out)
    args = [_p(_d(a)) for a in ins]
    getattr(lib(), name)(*args, _p(out))
    return out


def so3_exp(th):
    return _call_vec("orc_so3_exp", 4, th)


def so3_log(q_wxyz):
    return _call_vec("orc_so3_log", 3, q_wxyz)


def so3_ljac(th):
    return _call_vec("orc_so3_ljac", 9, th).reshape(3, 3)


def so3_ljacinv(th):
    return _call_vec("orc_so3_ljacinv", 9, th).reshape(3, 3)


def se3_exp(tau):
    return _call_vec("orc_se3_exp", 7, tau)


def se3_log(T):
    return _call_vec("orc_se3_log", 6, T)


def se3_compose(A, B):
    return _call_vec("orc_se3_compose", 7, A, B)


def se3_inverse(A):
    return _call_vec("orc_se3_inverse", 7, A)


def se3_adj(T):
    return _call_vec("orc_se3_adj", 36, T).reshape(6, 6)


def se3_rjac(tau):
    return _call_vec("orc_se3_rjac", 36, tau).reshape(6, 6)


def se3_rjacinv(tau):
    return _call_vec("orc_se3_rjacinv", 36, tau).reshape(6, 6)


def se3_ljacinv(tau):
    return _call_vec("orc_se3_ljacinv", 36, tau).reshape(6, 6)


# -------------------------------------------------------------------- model
def continuous_dynamics(mp, x, u, diffs=False):
    xdot = np.zeros(12)
    Jx = np.zeros((12, 12)) if diffs else None
    Ju = np.zeros((12, 4)) if diffs else None
    rc = lib().orc_continuous_dynamics(C.byref(mp), _p(_d(x)), _p(_d(u)), _p(xdot), _p(Jx), _p(Ju))
    if rc:
        raise RuntimeError("Inertia matrix is not positive definite!")
    return (xdot, Jx, Ju) if diffs else xdot


def discrete_dynamics(mp, x, u, dt, diffs=False):
    xn = np.zeros(13)
    Jx = np.zeros((12, 12)) if diffs else None
    Ju = np.zeros((12, 4)) if diffs else None
    rc = lib().orc_discrete_dynamics(C.byref(mp), _p(_d(x)), _p(_d(u)), C.c_double(dt), _p(xn),
                                     _p(Jx), _p(Ju))
    if rc:
        raise RuntimeError("Inertia matrix is not positive definite!")
    return (xn, Jx, Ju) if diffs else xn


def discrete_step(mp, integrator, x, u, dt, diffs=False):
    """orc_discrete_step: integrator 0 = explicit Euler (the reference), 1 = Runge-Kutta (extension)"""
    xn = np.zeros(13)
    Jx = np.zeros((12, 12)) if diffs else None
    Ju = np.zeros((12, 4)) if diffs else None
    rc = lib().orc_discrete_step(C.byref(mp), C.c_int(integrator), _p(_d(x)), _p(_d(u)), C.c_double(dt), _p(xn),
                                 _p(Jx), _p(Ju))
    if rc:
        raise RuntimeError("orc_discrete_step failed (%d)" % rc)
    return (xn, Jx, Ju) if diffs else xn


def state_add(x, tangent, diffs=False):
    out = np.zeros(13)
    Jl = np.zeros((12, 12)) if diffs else None
    Jr = np.zeros((12, 12)) if diffs else None
    lib().orc_state_add(_p(_d(x)), _p(_d(tangent)), _p(out), _p(Jl), _p(Jr))
    return (out, Jl, Jr) if diffs else out


def state_minus(lhs, rhs, diffs=False):
    out = np.zeros(12)
    Jl = np.zeros((12, 12)) if diffs else None
    Jr = np.zeros((12, 12)) if diffs else None
    lib().orc_state_minus(_p(_d(lhs)), _p(_d(rhs)), _p(out), _p(Jl), _p(Jr))
    return (out, Jl, Jr) if diffs else out


def euler_step(x, xdot, dt, diffs=False):
    out = np.zeros(13)
    Jl = np.zeros((12, 12)) if diffs else None
    Jr = np.zeros((12, 12)) if diffs else None
    lib().orc_euler_step(_p(_d(x)), _p(_d(xdot)), C.c_double(dt), _p(out), _p(Jl), _p(Jr))
    return (out, Jl, Jr) if diffs else out


def cost(Q, R, x, u, xd, ud, diffs=False):
    if not diffs:
        return lib().orc_cost(_p(_d(Q)), _p(_d(R)), _p(_d(x)), _p(_d(u)), _p(_d(xd)), _p(_d(ud)),
                              None, None, None, None, None)
    Cx, Cu = np.zeros(12), np.zeros(4)
    Cxx, Cuu, Cxu = np.zeros((12, 12)), np.zeros((4, 4)), np.zeros((12, 4))
    c = lib().orc_cost(_p(_d(Q)), _p(_d(R)), _p(_d(x)), _p(_d(u)), _p(_d(xd)), _p(_d(ud)),
                       _p(Cx), _p(Cu), _p(Cxx), _p(Cuu), _p(Cxu))
    return c, dict(x=Cx, u=Cu, xx=Cxx, uu=Cuu, xu=Cxu)


def ldlt4_solve(A, B):
    B = _d(B)
    B2 = B.reshape(4, -1)
    X = np.zeros_like(B2)
    lib().orc_ldlt4_solve(_p(_d(A)), _p(np.ascontiguousarray(B2)), C.c_int(B2.shape[1]), _p(X))
    return X.reshape(B.shape)


def gains_to_kK(gains):
    """(N,52) -> k (N,4), K (N,4,12)"""
    g = np.asarray(gains)
    return g[..., :4], np.swapaxes(g[..., 4:].reshape(g.shape[:-1] + (12, 4)), -1, -2)


def kK_to_gains(k, K):
    k = np.asarray(k, dtype=np.float64)
    K = np.asarray(K, dtype=np.float64)
    return np.concatenate([k, np.swapaxes(K, -1, -2).reshape(K.shape[:-2] + (48,))], axis=-1)


# ------------------------------------------------------------------- solver
class OracleSolver:
    """ILQR<QuadrotorModel> of the reference (ilqr.hh:25-206) on the CPU oracle."""

    def __init__(self, mp, Q, R, desired, dt, opt, library=None):
        self._L = library if library is not None else lib()  # (library: fast_library() for timing; the parity build otherwise)
        self.mp, self.opt, self.dt = mp, opt, dt
        self.desired = _d(desired).reshape(-1, 18)
        self._h = C.c_void_p()
        rc = self._L.orc_solver_create(C.byref(mp), _p(_d(Q)), _p(_d(R)), _p(self.desired),
                                     C.c_int(self.desired.shape[0]), C.c_double(dt),
                                     C.byref(opt), C.byref(self._h))
        if rc == ERR_BAD_INERTIA:
            raise RuntimeError("Inertia matrix is not positive definite!")
        if rc:
            raise ValueError(f"orc_solver_create failed: {rc}")

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_solver_destroy(self._h)
            self._h = None

    def flavour(self):
        return self._L.orc_build_flavour().decode()

    @staticmethod
    def _check(rc):
        if rc == ERR_LENGTH_MISMATCH:
            raise IndexError("trajectory longer than desired trajectory")
        if rc:
            raise ValueError(f"oracle error {rc}")

    def cost_trajectory(self, traj):
        traj = _d(traj).reshape(-1, 18)
        c = C.c_double()
        self._check(self._L.orc_cost_trajectory(self._h, _p(traj), C.c_int(len(traj)), C.byref(c)))
        return c.value

    def backwards_pass(self, traj):
        traj = _d(traj).reshape(-1, 18)
        gains = np.zeros((len(traj), 52))
        terms = np.zeros(2)
        self._check(self._L.orc_backwards_pass(self._h, _p(traj), C.c_int(len(traj)), _p(gains),
                                             _p(terms)))
        return gains, terms

    def backwards_pass_reg(self, traj, mu):
        """Extension (not in the reference): backward pass with mu on the diagonal of Q_uu."""
        traj = _d(traj).reshape(-1, 18)
        gains = np.zeros((len(traj), 52))
        terms = np.zeros(2)
        self._check(self._L.orc_backwards_pass_reg(self._h, _p(traj), C.c_int(len(traj)), C.c_double(mu),
                                                 _p(gains), _p(terms)))
        return gains, terms

    def set_regularisation(self, mu_init, mu_factor=10.0, mu_max=1e6):
        """Extension (not in the reference): Levenberg-Marquardt restarts in solve / solve_batch."""
        self._check(self._L.orc_set_regularisation(self._h, C.c_double(mu_init), C.c_double(mu_factor),
                                                 C.c_double(mu_max)))

    def set_integrator(self, integrator):
        """Extension (not in the reference's executed code): 1 = the Runge-Kutta step of quadrotor_model.cc:51-63."""
        self._check(self._L.orc_set_integrator(self._h, C.c_int(integrator)))

    def set_recursion(self, mode):
        """Extension (orc_set_recursion): 0 = ilqr.hh:132-133 as written (default), 1 = the substituted, symmetrised form
        V_x = Q_x + K^T Q_u, V_xx = sym(Q_xx + Q_xu K), k^T Q_uu k = -Q_u^T k -- stable at 200 / 500 knots, where the reference's own
        form is rounding noise; the comparand of the full-size tests of BASELINE.json configs[2] and configs[4]."""
        self._check(self._L.orc_set_recursion(self._h, C.c_int(mode)))

    def forward_sim(self, traj, gains, alpha=1.0):
        traj = _d(traj).reshape(-1, 18)
        gains = _d(gains).reshape(-1, 52)
        out = np.zeros_like(traj)
        self._check(self._L.orc_forward_sim(self._h, _p(traj), C.c_int(len(traj)), _p(gains),
                                          C.c_double(alpha), _p(out)))
        return out

    def line_search(self, traj, cost, gains, terms):
        traj = _d(traj).reshape(-1, 18)
        gains = _d(gains).reshape(-1, 52)
        out = np.zeros_like(traj)
        c, step, trials = C.c_double(), C.c_double(), C.c_int()
        st = self._L.orc_line_search(self._h, _p(traj), C.c_int(len(traj)), C.c_double(cost),
                                   _p(gains), _p(_d(terms)), _p(out), C.byref(c), C.byref(step),
                                   C.byref(trials))
        if st < 0:
            self._check(-st)
        return dict(status=st, traj=out, cost=c.value, step=step.value, trials=trials.value)

    def solve(self, init, debug=False, cap=None):
        init = _d(init).reshape(-1, 18)
        n = len(init)
        cap = int(self.opt.max_iters) + 1 if cap is None else cap
        out = np.zeros_like(init)
        hist = np.zeros(cap)
        dbg = np.zeros((cap, n, 18)) if debug else None
        c = C.c_double()
        st, it, nb, nf, nh = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._check(self._L.orc_solve(self._h, _p(init), C.c_int(n), _p(out), C.byref(c),
                                    C.byref(st), C.byref(it), C.byref(nb), C.byref(nf), _p(hist),
                                    _p(dbg), C.c_int(cap), C.byref(nh)))
        k = min(nh.value, cap)
        return dict(traj=out, cost=c.value, status=st.value, iters=it.value, n_bwd=nb.value,
                    n_fwd=nf.value, cost_hist=hist[:k].copy(),
                    debug_trajs=dbg[:k].copy() if debug else None)

    def solve_decisions(self, init):
        """solve() plus every comparison its control flow took (ilqr.hh:66, :186, :82) with the margin by which it came out
        the way it did, as a cost difference over |cost|: list of dicts in evaluation order."""
        init = _d(init).reshape(-1, 18)
        n = len(init)
        cap = int(self.opt.max_iters) + 1
        dcap = (cap + 1) * (2 + max(int(self.opt.ls_max_iters), 1))
        out = np.zeros_like(init)
        hist = np.zeros(cap)
        dec = (Decision * dcap)()
        c = C.c_double()
        st, it, nb, nf, nh, nd = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._check(self._L.orc_solve_decisions(self._h, _p(init), C.c_int(n), _p(out), C.byref(c), C.byref(st), C.byref(it),
                                              C.byref(nb), C.byref(nf), _p(hist), C.c_int(cap), C.byref(nh), dec,
                                              C.c_int(dcap), C.byref(nd)))
        assert nd.value <= dcap
        decisions = [dict(kind=d.kind, iter=d.iter, trial=d.trial, result=bool(d.result), n_bwd=d.n_bwd, n_fwd=d.n_fwd,
                          lhs=d.lhs, rhs=d.rhs, margin=d.margin) for d in dec[:nd.value]]
        return dict(traj=out, cost=c.value, status=st.value, iters=it.value, n_bwd=nb.value, n_fwd=nf.value,
                    cost_hist=hist[:min(nh.value, cap)].copy(), decisions=decisions)

    def solve_batch(self, init, n_threads=1):
        init = _d(init)
        B, n = init.shape[0], init.shape[1]
        out = np.zeros_like(init)
        cost = np.zeros(B)
        st, it, nb, nf = (np.zeros(B, dtype=np.int32) for _ in range(4))
        self._check(self._L.orc_solve_batch(self._h, _p(init), C.c_int(B), C.c_int(n), _p(out),
                                          _p(cost), _ip(st), _ip(it), _ip(nb), _ip(nf),
                                          C.c_int(n_threads)))
        return dict(traj=out, cost=cost, status=st, iters=it, n_bwd=nb, n_fwd=nf)
