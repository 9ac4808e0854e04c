"""The lane-local device functions (quadrotorilqr_amd/csrc/se3_math.h) and the backward pass's
operand layout (backward_layout.h), compiled for the host by tests/host_harness.cpp and checked
against the oracle.  This is test scaffolding: the shipped library has no host path.

Tolerances: per-function 1e-12 relative, per-pass 1e-10 relative.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc
from quadrotorilqr_amd import problems as pb

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hh():
    so = os.path.join(HERE, "libhost_harness.so")
    src = os.path.join(HERE, "host_harness.cpp")
    hdrs = [os.path.join(HERE, "..", "quadrotorilqr_amd", "csrc", h)
            for h in ("se3_math.h", "backward_layout.h", "host_model.h", "rollout16.h")]
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src, "-lm"])
    return C.CDLL(so)


def P(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def consts(hh, model, Q, R, dt):
    buf = np.zeros(hh.hh_consts_size() // 8)
    rc = hh.hh_make_consts(C.c_double(model["mass_kg"]), P(np.ascontiguousarray(model["inertia"], dtype=float)),
                           C.c_double(model["arm_length_m"]), C.c_double(model["torque_to_thrust_ratio_m"]),
                           C.c_double(model["g_mpss"]), P(np.ascontiguousarray(Q, dtype=float)),
                           P(np.ascontiguousarray(R, dtype=float)), C.c_double(dt), P(buf))
    assert rc == 0
    return buf


def layout(hh, c, force_general=False):
    lay = np.zeros(6, dtype=np.int32)
    hh.hh_layout(P(c), C.c_int(int(force_general)), lay.ctypes.data_as(C.POINTER(C.c_int)))
    return lay


def IP(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def random_problem(seed, n=12, dense=False):
    r = np.random.default_rng(seed)
    A = r.uniform(-1, 1, (3, 3))
    model = dict(mass_kg=1.3, inertia=A @ A.T + 3 * np.eye(3), arm_length_m=0.7,
                 torque_to_thrust_ratio_m=0.2, g_mpss=9.81)
    if dense == "sym":
        Q = r.uniform(-1, 1, (12, 12))
        Q = Q @ Q.T + 12 * np.eye(12)                                      # dense, symmetric
        R = r.uniform(-0.3, 0.3, (4, 4))
        R = R + R.T + 2 * np.eye(4)
    elif dense:
        Q = r.uniform(-1, 1, (12, 12))
        Q = Q @ Q.T + 12 * np.eye(12) + 0.3 * r.uniform(-1, 1, (12, 12))  # not symmetric
        R = r.uniform(-0.3, 0.3, (4, 4)) + 2 * np.eye(4)                   # not symmetric
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
    return model, Q, R, rand_traj(), rand_traj()


@pytest.mark.parametrize("seed,dense", [(1, False), (2, True), (3, True), (9, "sym")])
def test_knot_records_match_oracle(hh, seed, dense):
    model, Q, R, traj, desired = random_problem(seed, dense=dense)
    dt = 0.1
    c = consts(hh, model, Q, R, dt)
    n = len(traj)
    lay = layout(hh, c)
    assert list(lay[:2]) == {False: [1, 1], True: [0, 0], "sym": [1, 0]}[dense]
    assert lay[5] == {False: 92, True: 216, "sym": 128}[dense]
    assert hh.hh_check_operand_tables(P(c), IP(lay)) == 0
    lin = np.zeros((n, lay[5]))
    hh.hh_linearize(P(c), IP(lay), P(traj), P(desired), C.c_int(n), P(lin))
    mp = orc.model_params(**model)
    for i in range(n):
        _, Jx, Ju = orc.discrete_dynamics(mp, traj[i, 1:14], traj[i, 14:18], dt, diffs=True)
        jx, ju = np.zeros((12, 12)), np.zeros((12, 4))
        hh.hh_dense_jacobians(P(c), P(lin[i]), P(jx), P(ju))
        np.testing.assert_allclose(jx, Jx, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(ju, Ju, rtol=1e-12, atol=1e-15)
        cost, D = orc.cost(Q, R, traj[i, 1:14], traj[i, 14:18], desired[i, 1:14], desired[i, 14:18], diffs=True)
        np.testing.assert_allclose(lin[i, lay[4]], cost, rtol=1e-13)
        scale = np.abs(D["xx"]).max()
        cxx = np.zeros((12, 12))
        hh.hh_dense_cxx(P(c), IP(lay), P(lin[i]), P(cxx))
        np.testing.assert_allclose(cxx, D["xx"], rtol=1e-11, atol=1e-12 * scale)
        np.testing.assert_allclose(lin[i, lay[3]:lay[3] + 12], D["x"], rtol=1e-11, atol=1e-12 * np.abs(D["x"]).max())
        np.testing.assert_allclose(lin[i, lay[3] + 12:lay[3] + 16], D["u"], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("seed,dense", [(4, False), (5, True), (6, "sym")])
def test_tiled_placement_of_the_records_is_a_permutation_of_the_plain_one(hh, seed, dense):
    """rec_base / rec_elem and the paired stores of TiledRecWriter (what k_linearize writes for k_backward4, k_backward2
    and k_solve4): the same entries as the plain records, bit for bit, each in a place of its own (nothing else of the
    buffer is touched), for batches that do not fill their last tile"""
    model, Q, R, _, desired = random_problem(seed, n=5, dense=dense)
    c = consts(hh, model, Q, R, 0.1)
    lay = layout(hh, c)
    B, n, stride = 7, 5, int(lay[5])
    trajs = np.stack([random_problem(100 * seed + b, n=n, dense=dense)[3] for b in range(B)])
    plain = np.zeros((B, n, stride))
    for b in range(B):
        hh.hh_linearize(P(c), IP(lay), P(trajs[b]), P(desired), C.c_int(n), P(plain[b]))
    hh.hh_rec_count.restype = C.c_long
    count = hh.hh_rec_count(C.c_int(B), C.c_int(n), C.c_int(stride))
    tiled = np.full(count, np.nan)
    back = np.zeros((B, n, stride))
    tile = hh.hh_linearize_tiled(P(c), IP(lay), P(trajs), P(desired), C.c_int(B), C.c_int(n), P(tiled), P(back))
    assert tile in (4, 8, 16, 32, 64)
    used = lay[4] + 1  # entries a record holds (the stride is rounded up to an even number)
    np.testing.assert_array_equal(back[..., :used], plain[..., :used])
    assert np.isfinite(tiled).sum() == B * n * used  # every entry written once, to a place of its own


def test_knot_records_across_series_branches(hh):
    # the linearisation's Jacobian coefficients switch between manif's small-angle constants
    # (theta^2 <= 1e-10), power series (theta^2 <= 0.25 / 0.26) and the closed forms: sweep the step
    # rotation dt*omega and the orientation error across all of them
    model, Q, R = pb.MODEL_D, pb.Q_DEMO, pb.R_DEMO
    dt = 0.1
    c = consts(hh, model, Q, R, dt)
    lay = layout(hh, c)
    mp = orc.model_params(**model)
    r = np.random.default_rng(5)
    angles = [0.0, 1e-7, 9e-6, 1.1e-5, 1e-3, 0.05, 0.3, 0.499, 0.501, 0.509, 0.511, 0.8, 2.0, 3.0]
    for ang in angles:
        ax = r.normal(size=3); ax /= np.linalg.norm(ax)
        ax2 = r.normal(size=3); ax2 /= np.linalg.norm(ax2)
        des = np.zeros((1, 18)); tr = np.zeros((1, 18))
        des[0, 1:8] = orc.se3_exp(np.concatenate([r.uniform(-1, 1, 3), r.uniform(-1, 1, 3)]))
        des[0, 8:14] = r.uniform(-1, 1, 6)
        des[0, 14:18] = r.uniform(0, 5, 4)
        err = np.concatenate([r.uniform(-1, 1, 3), ang * ax])
        tr[0, 1:8] = orc.se3_compose(des[0, 1:8], orc.se3_exp(err))
        tr[0, 8:11] = r.uniform(-2, 2, 3)
        tr[0, 11:14] = ang / dt * ax2
        tr[0, 14:18] = r.uniform(0, 5, 4)
        lin = np.zeros((1, lay[5]))
        hh.hh_linearize(P(c), IP(lay), P(tr), P(des), C.c_int(1), P(lin))
        _, Jx, Ju = orc.discrete_dynamics(mp, tr[0, 1:14], tr[0, 14:18], dt, diffs=True)
        jx, ju = np.zeros((12, 12)), np.zeros((12, 4))
        hh.hh_dense_jacobians(P(c), P(lin[0]), P(jx), P(ju))
        # Just above theta^2 = 1e-10 the reference's closed forms cancel: (1 - cos th)/th^2 carries ~1e-16/th^2
        # relative error and the Q-block's (1 - th^2/2 - cos th)/th^4 an ABSOLUTE error ~1e-16/th^4 (thousands at
        # th = 1.1e-5), which the th^2 |rho| factors it multiplies reduce to ~1e-16/th^2 in the Jacobian.  The
        # series used here do not cancel, so that noise of the reference is the tolerance.
        noise = 1e-12 + (2e-16 / ang ** 2 if ang > 1e-5 else 0.0)
        np.testing.assert_allclose(jx, Jx, rtol=1e-12, atol=noise, err_msg=f"angle {ang}")
        cost, D = orc.cost(Q, R, tr[0, 1:14], tr[0, 14:18], des[0, 1:14], des[0, 14:18], diffs=True)
        np.testing.assert_allclose(lin[0, lay[4]], cost, rtol=1e-12, atol=1e-25)
        cxx = np.zeros((12, 12))
        hh.hh_dense_cxx(P(c), IP(lay), P(lin[0]), P(cxx))
        scale = np.abs(D["xx"]).max()
        np.testing.assert_allclose(cxx, D["xx"], rtol=1e-11, atol=noise * scale, err_msg=f"angle {ang}")
        np.testing.assert_allclose(lin[0, lay[3]:lay[3] + 12], D["x"], rtol=1e-11,
                                   atol=noise * max(np.abs(D["x"]).max(), 1e-300), err_msg=f"angle {ang}")


def test_knot_records_at_singular_points(hh):
    # desired roll = pi (w = 0 branch of Log), zero rotation rate (small-angle Exp), zero error
    model, Q, R = pb.MODEL_D, pb.Q_DEMO, pb.R_DEMO
    desired = pb.box_climb_desired(4.0)
    traj = desired.copy()
    c = consts(hh, model, Q, R, 0.1)
    n = len(traj)
    lay = layout(hh, c)
    lin = np.zeros((n, lay[5]))
    hh.hh_linearize(P(c), IP(lay), P(traj), P(desired), C.c_int(n), P(lin))
    assert np.all(np.isfinite(lin))
    np.testing.assert_array_equal(lin[:, lay[4]], 0.0)  # cost_test.cc:27-39
    mp = orc.model_params(**model)
    for i in (0, 15, 39):
        _, Jx, _ = orc.discrete_dynamics(mp, traj[i, 1:14], traj[i, 14:18], 0.1, diffs=True)
        jx, ju = np.zeros((12, 12)), np.zeros((12, 4))
        hh.hh_dense_jacobians(P(c), P(lin[i]), P(jx), P(ju))
        np.testing.assert_allclose(jx, Jx, rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize("seed,dense", [(4, False), (5, True)])
def test_rollout_matches_oracle(hh, seed, dense):
    model, Q, R, traj, desired = random_problem(seed, n=30, dense=dense)
    r = np.random.default_rng(seed)
    gains = 0.05 * r.uniform(-1, 1, (30, 52))
    c = consts(hh, model, Q, R, 0.1)
    s = orc.OracleSolver(orc.model_params(**model), Q, R, desired, 0.1, orc.options())
    for alpha in (1.0, 0.25):
        out = np.zeros_like(traj)
        hh.hh_rollout(P(c), P(traj), P(gains), C.c_double(alpha), P(out), C.c_int(30))
        ref = s.forward_sim(traj, gains, alpha)
        np.testing.assert_allclose(out, ref, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("seed,dense,sym", [(6, False, 1), (6, False, 0), (7, True, 0), (8, True, 0), (10, "sym", 1),
                                             (10, "sym", 0)])
def test_backward_dataflow_matches_oracle(hh, seed, dense, sym):
    """k_backward re-enacted on the CPU with the documented v_mfma_f64_16x16x4_f64 lane maps:
    proves the operand tables, the accumulator->operand hand-off and the gain layout.  sym = 1 is the
    instantiation for symmetric weights (V_x, V_xx, k^T Quu k in their Q_xu forms), sym = 0 the general one."""
    model, Q, R, traj, desired = random_problem(seed, n=15, dense=dense)
    c = consts(hh, model, Q, R, 0.1)
    n = len(traj)
    lay = layout(hh, c)
    lin = np.zeros((n, lay[5]))
    hh.hh_linearize(P(c), IP(lay), P(traj), P(desired), C.c_int(n), P(lin))
    gains, terms = np.zeros((n, 52)), np.zeros(2)
    hh.hh_backward_emulated(P(c), IP(lay), P(lin), C.c_int(n), P(gains), P(terms), C.c_int(sym))
    s = orc.OracleSolver(orc.model_params(**model), Q, R, desired, 0.1, orc.options())
    g_ref, t_ref = s.backwards_pass(traj)
    np.testing.assert_allclose(terms, t_ref, rtol=1e-10)
    np.testing.assert_allclose(gains, g_ref, rtol=1e-9, atol=1e-10 * np.abs(g_ref).max())


def _angles():
    # straddle every branch threshold of the fast arithmetic: manif's 1e-10 switches, the series
    # limits (s^2 <= 0.0625 <=> angle <= 0.5054; theta^2 <= 0.26 / 0.25), the w < 0 log branch, pi
    out = [0.0, 1e-7, 0.99e-5, 1.01e-5, 1e-3, 0.1, 0.4999, 0.5001, 0.505, 0.506, 0.5098, 0.5100, 0.51, 1.0, 2.0,
           3.0, np.pi - 1e-6, np.pi, 3.2, 4.0, 6.0]
    return out


def test_fast_rminus_matches_oracle_across_branches(hh):
    r = np.random.default_rng(0)
    for ang in _angles():
        for _ in range(4):
            ax = r.standard_normal(3)
            ax /= np.linalg.norm(ax)
            X = orc.se3_exp(np.concatenate([r.uniform(-2, 2, 3), r.uniform(-1.5, 1.5, 3)]))
            Y = orc.se3_compose(X, orc.se3_exp(np.concatenate([r.uniform(-1, 1, 3), ax * ang])))
            ref = orc.se3_log(orc.se3_compose(orc.se3_inverse(X), Y))
            got = np.zeros(6)
            hh.hh_rminus_fast(P(Y), P(X), P(got))
            # (1 + cos)/(2 theta sin) is 0/0 at pi: manif's own formula loses digits there
            tol = 1e-8 if abs(ang - np.pi) < 0.2 else 2e-13
            np.testing.assert_allclose(got, ref, rtol=1e-12, atol=tol * max(1.0, np.abs(ref).max()), err_msg=str(ang))
    X = orc.se3_exp(np.array([0.3, -1.0, 2.0, 0.7, -0.2, 1.1]))
    got = np.ones(6)
    hh.hh_rminus_fast(P(X), P(X), P(got))
    np.testing.assert_array_equal(got, 0.0)  # x (-) x == 0 exactly (cost_test.cc:27-39)


def test_fast_rplus_matches_oracle_across_branches(hh):
    r = np.random.default_rng(1)
    for ang in _angles():
        for _ in range(4):
            ax = r.standard_normal(3)
            ax /= np.linalg.norm(ax)
            X = orc.se3_exp(np.concatenate([r.uniform(-2, 2, 3), r.uniform(-1.5, 1.5, 3)]))
            tau = np.concatenate([r.uniform(-1, 1, 3), ax * ang])
            ref = orc.se3_compose(X, orc.se3_exp(tau))
            got = np.zeros(7)
            hh.hh_rplus_fast(P(X), P(tau), P(got))
            # manif's closed forms (1 - cos)/th^2 and (th - sin)/th^3 cancel catastrophically just above
            # their 1e-10 switch (relative error ~ 1e-16/th^2); the series used on the device do not
            tol = 2e-13 + (4e-16 / ang if ang > 1e-5 else 0.0)
            np.testing.assert_allclose(got, ref, rtol=0, atol=tol * max(1.0, np.abs(ref).max()), err_msg=str(ang))


def test_rollout_near_nominal_matches_oracle(hh):
    """rollouts of a converging solve: small relative rotations, i.e. the series branches"""
    cfg = pb.config2(B=2, N=40)
    c = consts(hh, cfg["model"], cfg["Q"], cfg["R"], cfg["dt"])
    s = orc.OracleSolver(orc.model_params(**cfg["model"]), cfg["Q"], cfg["R"], cfg["desired"], cfg["dt"],
                         orc.options(**cfg["options"]))
    traj = cfg["init"][1]
    for it in range(4):
        gains, _ = s.backwards_pass(traj)
        for alpha in (1.0, 0.5):
            out = np.zeros_like(traj)
            hh.hh_rollout(P(c), P(traj), P(gains), C.c_double(alpha), P(out), C.c_int(len(traj)))
            ref = s.forward_sim(traj, gains, alpha)
            np.testing.assert_allclose(out, ref, rtol=1e-10, atol=1e-11)
        traj = s.forward_sim(traj, gains, 1.0)


def test_tiled_layout_rollout_equals_plain_layout(hh):
    """the [tile][knot][pair][lane][2] indexing used on the device, exercised on the CPU"""
    for B, n in [(1, 5), (64, 3), (70, 9), (130, 2)]:
        cfg = pb.config2(B=B, N=n, seed=11)
        c = consts(hh, cfg["model"], cfg["Q"], cfg["R"], cfg["dt"])
        r = np.random.default_rng(B)
        traj = cfg["init"] + 0.1 * r.standard_normal(cfg["init"].shape) * (np.arange(18) >= 8)
        gains = 0.05 * r.uniform(-1, 1, (B, n, 52))
        alpha = 0.5 ** r.integers(0, 3, B).astype(float)
        out_t = np.zeros_like(traj)
        hh.hh_rollout_tiled(P(c), P(traj), P(gains), P(alpha), P(out_t), C.c_int(B), C.c_int(n))
        for b in range(B):
            out = np.zeros((n, 18))
            hh.hh_rollout(P(c), P(np.ascontiguousarray(traj[b])), P(np.ascontiguousarray(gains[b])),
                          C.c_double(alpha[b]), P(out), C.c_int(n))
            np.testing.assert_array_equal(out_t[b], out)


# ------------------------------------------------------------------ sixteen lanes per trajectory (rollout16.h)
def _random_cfg16(seed, n, dense=False):
    """model, weights and four random trajectories (random_problem's generator, four times)"""
    model, Q, R, t0, desired = random_problem(seed, n=n, dense=dense)
    trajs = np.stack([t0] + [random_problem(seed + 100 * k, n=n, dense=dense)[3] for k in (1, 2, 3)])
    cfg = dict(model=model, Q=Q, R=R, dt=0.1, desired=desired, options=dict(pb.OPTIONS_DEMO, populate_debug=False))
    return cfg, trajs


def _rollout16(hh, c, trajs, gains, alpha, ops_knot=-1):
    """four trajectories through the CPU re-enactment of k_rollout16's two wavefronts"""
    trajs, gains, alpha = (np.ascontiguousarray(a, dtype=float) for a in (trajs, gains, alpha))
    out = np.zeros_like(trajs)
    ops = np.zeros((13, 64))
    hh.hh_rollout16(P(c), P(trajs), P(gains), P(alpha), P(out), C.c_int(trajs.shape[1]), C.c_int(ops_knot), P(ops))
    return out, ops


def _pose_close(a, b, atol):
    """knots compared as the reference's tests do (ilqr_test.cc:38-53): poses up to the quaternion's sign"""
    a = a.copy()
    flip = np.sum(a[..., 4:8] * b[..., 4:8], axis=-1) < 0
    a[flip, 4:8] *= -1
    np.testing.assert_allclose(a, b, rtol=1e-10, atol=atol)


def test_rollout16_operand_registers(hh):
    """what wavefront P prepares for a knot, against the definitions: thirteen registers shared by quad -- the gain columns
    in Q3 of registers 0..11, with the rows of R_n^T in Q0 of registers 0..2 and the left-multiplication matrix of conj(q_n)
    in Q0 of registers 3..6; register 12: nominal linear velocity (Q0), nominal translation (Q1), nominal angular velocity
    (Q2), u_nom + alpha k (Q3); zeros in every lane that owns nothing"""
    cfg, trajs = _random_cfg16(41, 5)
    c = consts(hh, cfg["model"], cfg["Q"], cfg["R"], cfg["dt"])
    r = np.random.default_rng(5)
    tr = trajs[:4]
    gains = r.uniform(-1, 1, (4, 5, 52))
    alpha = np.array([1.0, 0.5, 0.25, 0.125])
    _, ops = _rollout16(hh, c, tr, gains, alpha, ops_knot=2)
    assert ops.shape[0] == 13
    for row in range(4):
        p = tr[row][2]
        w, x, y, z = p[4:8]
        Rn = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                       [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                       [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        L = np.array([[w, z, -y, -x], [-z, w, x, -y], [y, -x, w, -z], [x, y, z, w]])  # conj(q_n) (x) . on (x, y, z, w)
        k, K = orc.gains_to_kK(gains[row][2:3])
        for l in range(16):
            lane, q, j = 16 * row + l, l >> 2, l & 3
            for cc in range(12):
                if q == 3:
                    want = K[0][j, cc]
                elif q == 0 and cc < 3:
                    want = Rn[cc, j] if j < 3 else 0.0
                elif q == 0 and cc < 7:
                    want = L[j, cc - 3]
                else:
                    want = 0.0
                np.testing.assert_allclose(ops[cc, lane], want, atol=1e-15)
            want = [p[8 + j] if j < 3 else 0.0, p[1 + j] if j < 3 else 0.0, p[11 + j] if j < 3 else 0.0, p[14 + j] + alpha[row] * k[0][j]][q]
            np.testing.assert_allclose(ops[12, lane], want, rtol=1e-15)


@pytest.mark.parametrize("seed,dense", [(51, False), (52, True)])
def test_rollout16_matches_oracle_on_random_trajectories(hh, seed, dense):
    """random nominal trajectories (rotations up to 1.5 rad per axis away from anything: the closed-form branches of
    Log and of its Jacobian), small random gains, four different step sizes in the four rows"""
    cfg, trajs = _random_cfg16(seed, 14, dense)
    c = consts(hh, cfg["model"], cfg["Q"], cfg["R"], cfg["dt"])
    ref = orc.OracleSolver(orc.model_params(**cfg["model"]), cfg["Q"], cfg["R"], cfg["desired"], cfg["dt"],
                           orc.options(**cfg["options"]))
    r = np.random.default_rng(seed)
    tr = trajs[:4]
    gains = 0.05 * r.uniform(-1, 1, (4, 14, 52))
    alpha = np.array([1.0, 0.5, 0.25, 1.0])
    out, _ = _rollout16(hh, c, tr, gains, alpha)
    for b in range(4):
        _pose_close(out[b], ref.forward_sim(tr[b], gains[b], alpha[b]), atol=1e-10)
        np.testing.assert_array_equal(out[b][:, 0], tr[b][:, 0])          # time_s passes through
        np.testing.assert_array_equal(out[b][0, 1:14], tr[b][0, 1:14])    # knot 0 is the input's state


def test_rollout16_matches_oracle_along_a_converging_solve(hh):
    """the series branches: rollouts of a converging solve (small relative rotations), four trajectories at once"""
    cfg = pb.config2(B=4, N=40)
    c = consts(hh, cfg["model"], cfg["Q"], cfg["R"], cfg["dt"])
    s = orc.OracleSolver(orc.model_params(**cfg["model"]), cfg["Q"], cfg["R"], cfg["desired"], cfg["dt"],
                         orc.options(**cfg["options"]))
    trajs = cfg["init"].copy()
    for it in range(4):
        gains = np.stack([s.backwards_pass(t)[0] for t in trajs])
        for alpha in (np.array([1.0, 1.0, 1.0, 1.0]), np.array([0.5, 1.0, 0.25, 0.125])):
            out, _ = _rollout16(hh, c, trajs, gains, alpha)
            for b in range(4):
                np.testing.assert_allclose(out[b], s.forward_sim(trajs[b], gains[b], alpha[b]), rtol=1e-10, atol=1e-11)
        trajs = np.stack([s.forward_sim(trajs[b], gains[b], 1.0) for b in range(4)])


def test_rollout16_branches(hh):
    """(a) a rollout that stays ON the nominal trajectory (zero gains, nominal = a feasible rollout): relative rotation
    zero -> manif's small-angle branches of Log and of its Jacobian; zero angular velocity -> small-angle Exp;
    (b) fast spin (|dt omega| > 0.5 rad) -> the closed forms of Exp; (c) the demo's desired trajectory, whose roll
    reaches pi (w = 0 exactly: the atan2 branch of Log with w <= 0)"""
    cfg = pb.config2(B=4, N=12)
    c = consts(hh, cfg["model"], cfg["Q"], cfg["R"], cfg["dt"])
    s = orc.OracleSolver(orc.model_params(**cfg["model"]), cfg["Q"], cfg["R"], cfg["desired"], cfg["dt"],
                         orc.options(**cfg["options"]))
    zero = np.zeros((12, 52))
    init = cfg["init"].copy()
    init[0, 0, 11:14] = 0.0                      # (a) no angular velocity at all in row 0
    init[1, 0, 11:14] = [9.0, -4.0, 2.0]         # (b) |dt omega| ~ 1 rad
    feas = np.stack([s.forward_sim(init[b], zero, 1.0) for b in range(4)])   # feasible nominal trajectories
    out, _ = _rollout16(hh, c, feas, np.zeros((4, 12, 52)), np.ones(4))
    for b in range(4):
        np.testing.assert_allclose(out[b], s.forward_sim(feas[b], zero, 1.0), rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(out[b], feas[b], rtol=1e-9, atol=1e-10)   # it re-traces the nominal trajectory
    r = np.random.default_rng(3)
    gains = 0.05 * r.uniform(-1, 1, (4, 12, 52))
    out, _ = _rollout16(hh, c, init, gains, np.array([1.0, 1.0, 0.5, 0.5]))
    for b in range(4):
        np.testing.assert_allclose(out[b], s.forward_sim(init[b], gains[b], [1.0, 1.0, 0.5, 0.5][b]), rtol=1e-10, atol=1e-11)
    # (b') the three ranges of Exp in ONE wavefront and across their boundaries: |dt omega|^2 = 0.2 (eight-term series), 3 and 11.5
    # (sixteen-term series, round 6), 13 and 30 (closed forms)
    for x2s in ((0.2, 3.0, 11.5, 13.0), (0.26, 11.9, 12.1, 30.0), (6.0, 6.0, 6.0, 6.0)):
        spin = cfg["init"].copy()
        for b, x2 in enumerate(x2s):
            d = r.normal(size=3)
            spin[b, 0, 11:14] = d / np.linalg.norm(d) * np.sqrt(x2) / cfg["dt"]
        out, _ = _rollout16(hh, c, spin, np.zeros((4, 12, 52)), np.ones(4))
        for b in range(4):
            np.testing.assert_allclose(out[b], s.forward_sim(spin[b], zero, 1.0), rtol=1e-10, atol=1e-11)
    # (c) the demo's box climb, 40 knots, first iterations of its solve
    cfg1 = pb.config1(4.0)
    c1 = consts(hh, cfg1["model"], cfg1["Q"], cfg1["R"], cfg1["dt"])
    s1 = orc.OracleSolver(orc.model_params(**cfg1["model"]), cfg1["Q"], cfg1["R"], cfg1["desired"], cfg1["dt"],
                          orc.options(**cfg1["options"]))
    traj = cfg1["init"][0]
    for it in range(3):
        g, _ = s1.backwards_pass(traj)
        four = np.stack([traj] * 4)
        al = np.array([1.0, 0.5, 0.25, 0.125])
        out, _ = _rollout16(hh, c1, four, np.stack([g] * 4), al)
        for b in range(4):
            _pose_close(out[b], s1.forward_sim(traj, g, al[b]), atol=1e-9)
        traj = s1.forward_sim(traj, g, 1.0)


# ------------------------------------------------------------------ the Runge-Kutta extension (se3_math.h: rk4_step)
@pytest.mark.parametrize("seed,dense", [(21, False), (22, True), (23, "sym")])
def test_rk4_step_and_dense_records_match_oracle(hh, seed, dense):
    """The device code's Runge-Kutta step -- next state and M = [J_x | J_u] by the chain rule with the primitives' sparsity
    written out -- against the oracle's statement of it from the reference's dense primitives (orc_discrete_step, 1); then
    the knot records of the dense layout: M at the head, the cost entries of the block layouts behind it."""
    model, Q, R, traj, desired = random_problem(seed, dense=dense)
    dt = 0.1
    c = consts(hh, model, Q, R, dt)
    mp = orc.model_params(**model)
    n = len(traj)
    for i in range(n):
        xn, Jx, Ju = orc.discrete_step(mp, 1, traj[i, 1:14], traj[i, 14:18], dt, diffs=True)
        got, MU = np.zeros(13), np.zeros((12, 16))
        hh.hh_rk4_step(P(c), P(traj[i, 1:14].copy()), P(traj[i, 14:18].copy()), P(got), P(MU))
        np.testing.assert_allclose(got, xn, rtol=0, atol=1e-13)
        np.testing.assert_allclose(MU[:, :12], Jx, rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(MU[:, 12:], Ju, rtol=1e-11, atol=1e-13)
    lay0, lay = layout(hh, c), np.zeros(6, dtype=np.int32)
    hh.hh_layout_rk4(P(c), C.c_int(0), lay.ctypes.data_as(C.POINTER(C.c_int)))
    assert lay[2] == 192 and lay[5] == lay0[5] + 138 and list(lay[:2]) == list(lay0[:2])
    lin0, lin = np.zeros((n, lay0[5])), np.zeros((n, lay[5]))
    hh.hh_linearize(P(c), IP(lay0), P(traj), P(desired), C.c_int(n), P(lin0))
    hh.hh_linearize(P(c), IP(lay), P(traj), P(desired), C.c_int(n), P(lin))
    np.testing.assert_array_equal(lin[:, 192:lay[4] + 1], lin0[:, 54:lay0[4] + 1])   # the cost half is the Euler layout's, moved
    for i in range(n):
        _, Jx, Ju = orc.discrete_step(mp, 1, traj[i, 1:14], traj[i, 14:18], dt, diffs=True)
        jx, ju = np.zeros((12, 12)), np.zeros((12, 4))
        hh.hh_dense_jacobians_lay(P(c), IP(lay), P(lin[i]), P(jx), P(ju))
        np.testing.assert_allclose(jx, Jx, rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(ju, Ju, rtol=1e-11, atol=1e-13)
        cxx0, cxx = np.zeros((12, 12)), np.zeros((12, 12))
        hh.hh_dense_cxx(P(c), IP(lay0), P(lin0[i]), P(cxx0))
        hh.hh_dense_cxx(P(c), IP(lay), P(lin[i]), P(cxx))
        np.testing.assert_array_equal(cxx, cxx0)
    # the Euler layouts through the layout-aware accessor are what they were
    jx0, ju0, jx1, ju1 = np.zeros((12, 12)), np.zeros((12, 4)), np.zeros((12, 12)), np.zeros((12, 4))
    hh.hh_dense_jacobians(P(c), P(lin0[0]), P(jx0), P(ju0))
    hh.hh_dense_jacobians_lay(P(c), IP(lay0), P(lin0[0]), P(jx1), P(ju1))
    np.testing.assert_array_equal(jx0, jx1)
    np.testing.assert_array_equal(ju0, ju1)


def test_rk4_rollout_matches_oracle(hh):
    model, Q, R = pb.MODEL_D, pb.Q_DEMO, pb.R_DEMO
    cfg = pb.config2(B=3, N=30, seed=8)
    c = consts(hh, cfg["model"], cfg["Q"], cfg["R"], cfg["dt"])
    o = orc.OracleSolver(orc.model_params(**cfg["model"]), cfg["Q"], cfg["R"], cfg["desired"], cfg["dt"], orc.options(**cfg["options"]))
    o.set_integrator(1)
    for b in range(3):
        tr = cfg["init"][b]
        gains, _ = o.backwards_pass(tr)
        for alpha in (1.0, 0.25):
            ref = o.forward_sim(tr, gains, alpha)
            out = np.zeros_like(tr)
            hh.hh_rollout_rk4(P(c), P(tr.copy()), P(gains.copy()), C.c_double(alpha), P(out), C.c_int(len(tr)))
            np.testing.assert_allclose(out, ref, rtol=0, atol=1e-10)


def test_pivoted_ldlt_of_the_general_kernel_is_eigens(hh):
    """ldlt4_pivoted_solve (backward_layout.h: the factorisation of Q_uu in k_backward<false>, i.e. non-symmetric weights or
    force_general = 1) against the oracle's restatement of Eigen 3.4.0 LDLT<Matrix4d, Lower> (ilqr.hh:126-128): the same
    pivots, hence the same solution to rounding, on SPD, indefinite, tiny-leading-pivot, tied-diagonal and singular inputs;
    only the lower triangle is read."""
    from oracle import oracle as orc
    r = np.random.default_rng(11)
    cases = []
    for _ in range(40):
        A = r.uniform(-1, 1, (4, 4))
        cases.append(A @ A.T + 0.1 * np.eye(4))                      # SPD
        S = r.uniform(-1, 1, (4, 4))
        cases.append((S + S.T) / 2)                                    # symmetric indefinite, random pivot order
    for p in (1e-4, 1e-8, 1e-12, 0.0):
        R = np.eye(4)
        R[0, 0] = p
        R[0, 1] = R[1, 0] = 1.0
        cases.append(R)                                                 # tiny (or zero) leading entry: the pivot moves
    T = np.full((4, 4), 0.25) + np.diag([2.0, 2.0, 2.0, 2.0])
    cases.append(T)                                                     # ties on the diagonal: the first one wins
    # ties where Eigen's exchanges are NOT a stable sort of the diagonal: |d| = (2, 1, 2, 3) ends as rows (3, 2, 0, 1) -- the exchange of
    # rows 0 and 3 moves the first 2 behind the second one (round 5: the order is found from the diagonal alone, ldlt4_pivot_order)
    for dg in ([2.0, 1.0, 2.0, 3.0], [1.0, 3.0, 3.0, 1.0], [2.0, 2.0, 1.0, 2.0], [-2.0, 2.0, 1.0, -2.0], [1.0, 1.0, 2.0, 2.0]):
        S = r.uniform(-0.3, 0.3, (4, 4))
        S = (S + S.T) / 2
        np.fill_diagonal(S, dg)
        cases.append(S)
    cases.append(np.diag([0.0, 3.0, 0.0, 1.0]))                         # zero pivots: Eigen's solve puts 0 there
    U = r.uniform(-1, 1, (4, 4))
    U[np.triu_indices(4, 1)] = 777.0                                    # garbage above the diagonal is never read
    cases.append(U + np.diag([3.0, 3.0, 3.0, 3.0]))
    for A in cases:
        A = np.ascontiguousarray(A, dtype=float)
        b = r.uniform(-1, 1, 4)
        x = np.zeros(4)
        hh.hh_ldlt4_pivoted_solve(P(A), P(b), P(x))
        ref = orc.ldlt4_solve(A, b)
        np.testing.assert_allclose(x, np.asarray(ref).reshape(4), rtol=1e-13, atol=1e-13 * max(1.0, np.abs(ref).max()))
        np.testing.assert_array_equal(x, np.asarray(ref).reshape(4))  # (the host build divides as the oracle does: the same operations, the same bits)
    # the documented case (tests/test_gpu_robustness.py): k = -du to rounding whatever the leading entry
    for p in (1e-4, 1e-8, 1e-12):
        R = np.eye(4)
        R[0, 0] = p
        R[0, 1] = R[1, 0] = 1.0
        du = r.uniform(-1, 1, 4)
        x = np.zeros(4)
        hh.hh_ldlt4_pivoted_solve(P(np.ascontiguousarray(2 * R)), P(np.ascontiguousarray(2 * R @ du)), P(x))
        assert np.abs(x - du).max() < 1e-13


def test_diagonal_weights_instantiation_gives_the_block_diagonal_one_bit_for_bit(hh):
    """linearize_cost<3> (round 3: Q exactly diagonal -- the reference's demo and tests -- so J^T Q is a row scaling) against
    linearize_cost<2> (Q symmetric with zero pose x velocity blocks) on the same diagonal Q: the sums of <2> add products with
    exact zeros to the products <3> keeps, so every entry of the record and the knot cost are the same numbers."""
    r = np.random.default_rng(8)
    model = dict(mass_kg=1.1, inertia=np.diag([1.0, 1.3, 0.8]), arm_length_m=0.7, torque_to_thrust_ratio_m=0.1, g_mpss=9.81)
    Q = np.diag(r.uniform(0.5, 120.0, 12))
    R = np.diag(r.uniform(0.5, 2.0, 4))
    c = consts(hh, model, Q, R, 0.1)
    hh.hh_linearize_cost_kind.restype = C.c_double
    stride = hh.hh_lin_stride()
    for trial in range(200):
        big = trial % 4 == 3   # large rotations: the closed-form branches of Log and of the Jacobian coefficients
        def knot():
            p = np.zeros(18)
            p[1:4] = r.uniform(-2, 2, 3)
            ax = r.normal(size=3); ax /= np.linalg.norm(ax)
            ang = r.uniform(-3.0, 3.0) if big else r.uniform(-0.4, 0.4)
            p[4] = np.cos(ang / 2); p[5:8] = np.sin(ang / 2) * ax
            p[8:14] = r.normal(size=6); p[14:18] = r.uniform(0, 5, 4)
            return p
        pt, pd = knot(), knot()
        rec2, rec3 = np.full(stride, 7.0), np.full(stride, 7.0)
        c2 = hh.hh_linearize_cost_kind(P(c), C.c_int(2), P(pt), P(pd), P(rec2))
        c3 = hh.hh_linearize_cost_kind(P(c), C.c_int(3), P(pt), P(pd), P(rec3))
        assert c2 == c3
        np.testing.assert_array_equal(rec3, rec2)   # (+0.0 == -0.0: the sign of a zero is the one thing that may differ)
