"""Workload definitions: the model sets and synthetic problem batches of BASELINE.md section 3.

Everything here is host-side NumPy (no oracle, no GPU).  Knot layout (N,18) is the
reference demo's IDX (quadrotor_ilqr.py:19-37):
    [time_s, tx,ty,tz, qw,qx,qy,qz, v_lin(3), v_ang(3), u0..u3]
"""
import numpy as np

PT = 18

# --------------------------------------------------------------------------- model sets
# Model D = the demo's constants (quadrotor_ilqr.py:286-292)
MODEL_D = dict(mass_kg=1.0, inertia=np.eye(3), arm_length_m=1.0, torque_to_thrust_ratio_m=0.0,
               g_mpss=9.81)
# Model A = D with a yaw-capable rotor: well-posed random-start family
MODEL_A = dict(MODEL_D, torque_to_thrust_ratio_m=0.1)

Q_DEMO = np.diag(np.concatenate((100 * np.ones(6), 1 * np.ones(6))))  # quadrotor_ilqr.py:291
R_DEMO = np.eye(4)                                                     # quadrotor_ilqr.py:292
DT_DEMO = 0.1                                                          # quadrotor_ilqr.py:257

# quadrotor_ilqr.py:272-284
OPTIONS_DEMO = dict(step_update=0.5, desired_reduction_frac=0.5, ls_max_iters=100,
                    rtol=1e-12, atol=1e-12, max_iters=100, populate_debug=True)


def hover_thrust(model):
    return model["mass_kg"] * model["g_mpss"] / 4.0


# --------------------------------------------------------------------------- SE(3) helpers
def _hat(a):
    z = np.zeros_like(a[..., 0])
    return np.stack([np.stack([z, -a[..., 2], a[..., 1]], -1),
                     np.stack([a[..., 2], z, -a[..., 0]], -1),
                     np.stack([-a[..., 1], a[..., 0], z], -1)], -2)


def se3_exp(tau):
    """Batched Exp: tau (...,6) = [rho ; theta] -> pose (...,7) = [t ; q(w,x,y,z)].
    Closed form (Sola et al. 2018, eqs. 172-174); used only to place synthetic start states."""
    tau = np.asarray(tau, dtype=np.float64)
    rho, th = tau[..., :3], tau[..., 3:]
    ang = np.linalg.norm(th, axis=-1)
    small = ang < 1e-8
    a = np.where(small, 1.0, ang)
    W = _hat(th)
    WW = W @ W
    c1 = np.where(small, 0.5, (1 - np.cos(a)) / a**2)[..., None, None]
    c2 = np.where(small, 1.0 / 6.0, (a - np.sin(a)) / a**3)[..., None, None]
    V = np.eye(3) + c1 * W + c2 * WW
    t = (V @ rho[..., None])[..., 0]
    half = 0.5 * ang
    sinc_half = np.where(small, 0.5, np.sin(half) / a)
    q = np.concatenate([np.cos(half)[..., None], th * sinc_half[..., None]], -1)
    return np.concatenate([t, q], -1)


# --------------------------------------------------------------------------- trajectories
def identity_trajectory(n, dt):
    """ilqr_test.cc:22-36: identity pose, zero velocity, zero control"""
    traj = np.zeros((n, PT))
    t = 0.0                          # the reference accumulates time_s += dt_s
    for i in range(n):
        traj[i, 0] = t
        t += dt
    traj[:, 4] = 1.0
    return traj


def box_climb_desired(horizon_s=4.0, dt=DT_DEMO, vel_mps=10.0):
    """The demo's desired trajectory (behaviour of quadrotor_ilqr.py:83-106, 256-270):
    four legs of a square, climbing 10/3 m per leg, roll stepping 0, pi/3, 2pi/3, pi;
    zero body velocity and zero control everywhere."""
    time_s = np.arange(0, horizon_s, dt)
    n = len(time_s)
    traj = np.zeros((n, PT))
    traj[:, 0] = time_s
    qh = horizon_s / 4.0
    for i, t in enumerate(time_s):
        if t < qh:
            p, roll = (vel_mps * t, 0.0, 0.0), 0.0
        elif t < 2.0 * qh:
            p, roll = (vel_mps * qh, vel_mps * (t - qh), 10.0 / 3.0), np.pi / 3.0
        elif t < 3.0 * qh:
            p, roll = (vel_mps * (3.0 * qh - t), vel_mps * qh, 20.0 / 3.0), 2.0 * np.pi / 3.0
        else:
            p, roll = (0.0, vel_mps * (4.0 * qh - t), 10.0), np.pi
        traj[i, 1:4] = p
        traj[i, 4] = np.cos(roll / 2.0)   # R.from_euler("xyz",[roll,0,0]) = rotation about x
        traj[i, 5] = np.sin(roll / 2.0)
    return traj


def hover_desired(n, dt=DT_DEMO, u_hover=0.0):
    traj = identity_trajectory(n, dt)
    traj[:, 0] = dt * np.arange(n)
    traj[:, 14:18] = u_hover
    return traj


# --------------------------------------------------------------------------- counter-based RNG
_M64 = (1 << 64) - 1


def _splitmix64(x):
    """vectorised splitmix64 finaliser on uint64 arrays"""
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(_M64)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(_M64)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(_M64)
    return z ^ (z >> np.uint64(31))


def counter_uniform(seed, b, comp):
    """U[0,1) keyed by (seed, problem index b, component) -- independent of batch size and
    of how the batch is sharded over ranks."""
    with np.errstate(over="ignore"):
        b = np.asarray(b, dtype=np.uint64)
        key = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + b)
        key = _splitmix64(key ^ (np.uint64(comp) * np.uint64(0xD6E8FEB86659FD93)))
    return (key >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def counter_normal(seed, b, comp):
    u1 = counter_uniform(seed, b, 2 * comp + 1000)
    u2 = counter_uniform(seed, b, 2 * comp + 1001)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def random_start_batch(b_index, desired, seed, pos_m=1.0, ang_rad=np.pi / 4, vel_sigma=0.5):
    """Initial trajectories for problems `b_index` (array of global problem ids):
    knot 0 = Exp([p ~ U(-1,1)^3 ; theta ~ U(-1,1)^3 * ang/sqrt(3)]) with body velocity
    ~ N(0, vel_sigma^2)^6; every other knot and every control = the desired trajectory's.
    """
    b_index = np.asarray(b_index, dtype=np.uint64)
    B = len(b_index)
    init = np.broadcast_to(desired, (B,) + desired.shape).copy()
    xi = np.stack([2.0 * counter_uniform(seed, b_index, c) - 1.0 for c in range(6)], -1)
    xi[:, :3] *= pos_m
    xi[:, 3:] *= ang_rad / np.sqrt(3.0)
    init[:, 0, 1:8] = se3_exp(xi)
    init[:, 0, 8:14] = vel_sigma * np.stack([counter_normal(seed, b_index, 10 + c) for c in range(6)], -1)
    return init


def config2(B=1024, N=100, seed=2, b0=0):
    """BASELINE.json configs[1]: model A, hover at identity, random SE(3) starts, fp64."""
    u = hover_thrust(MODEL_A)
    desired = hover_desired(N, DT_DEMO, u)
    init = random_start_batch(np.arange(b0, b0 + B), desired, seed)
    opts = dict(OPTIONS_DEMO, populate_debug=False)
    return dict(model=MODEL_A, Q=Q_DEMO, R=R_DEMO, dt=DT_DEMO, options=opts, desired=desired, init=init)


def config1(horizon_s=10.0):
    """BASELINE.json configs[0]: the demo problem at 100 knots, initial = desired."""
    desired = box_climb_desired(horizon_s)
    return dict(model=MODEL_D, Q=Q_DEMO, R=R_DEMO, dt=DT_DEMO, options=dict(OPTIONS_DEMO),
                desired=desired, init=desired[None].copy())


def config5(B=4096, N=500, seed=5):
    """BASELINE.json configs[4]: long-horizon stress.  Two halves that need two solver handles
    (different model and desired trajectory): (a) model A hover with random starts -- well posed;
    (b) the demo's box-climb stretched to N knots with random starts -- the unchecked first full step
    (ilqr.hh:71-73) usually diverges at this horizon, so line-search exhaustion and max-iteration
    exits are expected.  Graded on per-problem status and on parity where the oracle converges."""
    half = B // 2
    a = config2(B=half, N=N, seed=seed)
    desired = box_climb_desired(N * DT_DEMO)[:N]
    init = random_start_batch(np.arange(half, B), desired, seed)
    b = dict(model=MODEL_D, Q=Q_DEMO, R=R_DEMO, dt=DT_DEMO, options=dict(OPTIONS_DEMO, populate_debug=False),
             desired=desired, init=init)
    return a, b


def config3(B=8192, N=200, seed=3):
    """BASELINE.json configs[2]: as config 2 with 200 knots, fp32 (mixed precision: fp32 storage and
    lane-local arithmetic, fp64 Riccati recursion, fp64 cost sums / Armijo / convergence), and the
    convergence tolerances fp32 can reach: conv(1e-5, 1e-5, 100)."""
    cfg = config2(B=B, N=N, seed=seed)
    cfg["options"] = dict(cfg["options"], rtol=1e-5, atol=1e-5)
    return cfg
