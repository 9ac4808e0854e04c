"""ctypes binding of libquadrotor_ilqr.so (include/quadrotor_ilqr.h) and the host-side mirror
of the reference's ILQR<QuadrotorModel> interface (src/ilqr.hh:25-206): same method names,
argument meaning and error behaviour, batched.

There is no CPU fallback: if the HIP library is missing or no GPU is present every compute
entry point raises.
"""
import ctypes as C
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (QILQR_LIB: another build of the same library, for A/B measurements -- profiles/microbench)
LIB_PATH = os.environ.get("QILQR_LIB") or os.path.join(_HERE, "lib", "libquadrotor_ilqr.so")

KNOT = 18
GAIN = 52

OK = 0
ERR_BAD_INERTIA = 1
ERR_LENGTH_MISMATCH = 2
ERR_INVALID_ARG = 3
ERR_BAD_QUATERNION = 4
ERR_NO_DEVICE = 5
ERR_HIP = 6
ERR_LINE_SEARCH = 7

STATUS_CONVERGED_EXPECTED = 0
STATUS_CONVERGED = 1
STATUS_MAX_ITERS = 2
STATUS_LINE_SEARCH_FAILED = 3

# every symbol include/quadrotor_ilqr.h declares
EXPORTS = (
    "qilqr_create", "qilqr_create_sized", "qilqr_destroy", "qilqr_last_error", "qilqr_solve", "qilqr_solve_batch",
    "qilqr_solve_batch_device", "qilqr_cost_trajectory", "qilqr_backwards_pass", "qilqr_forward_sim",
    "qilqr_line_search", "qilqr_cost_history", "qilqr_profile_reset", "qilqr_profile_get", "qilqr_profile_mode", "qilqr_set_regularisation",
    "qilqr_set_integrator",
    "qilqr_device", "qilqr_stream", "qilqr_stream_wait_event", "qilqr_host_alloc", "qilqr_host_free",
    "qilqr_sharded_create", "qilqr_sharded_create_sized", "qilqr_sharded_create_mask", "qilqr_sharded_create_mask_sized", "qilqr_sharded_destroy", "qilqr_sharded_count", "qilqr_sharded_solver",
    "qilqr_shard_range", "qilqr_solve_batch_sharded",
    "qilqr_sharded_set_transport", "qilqr_sharded_transport", "qilqr_solve_batch_sharded_device", "qilqr_gather_schedule",
    "qilqr_abi_version", "qilqr_compaction_moves", "qilqr_describe",
)


class Model(C.Structure):
    _fields_ = [("mass_kg", C.c_double), ("inertia", C.c_double * 9), ("arm_length_m", C.c_double),
                ("torque_to_thrust_ratio_m", C.c_double), ("g_mpss", C.c_double)]


class Options(C.Structure):
    _fields_ = [("step_update", C.c_double), ("desired_reduction_frac", C.c_double),
                ("ls_max_iters", C.c_int32), ("rtol", C.c_double), ("atol", C.c_double),
                ("max_iters", C.c_double), ("populate_debug", C.c_int32)]


class DeviceConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("profile", C.c_int32), ("sync_every", C.c_int32),
                ("force_general", C.c_int32), ("single_wave_rollout", C.c_int32), ("precision", C.c_int32),
                ("streams", C.c_int32), ("persistent", C.c_int32), ("compaction", C.c_int32),
                # ABI version 7: the A/B switches that were environment variables
                ("round_launch", C.c_int32), ("rounds_per_launch", C.c_int32), ("fuse_in_flight", C.c_int32), ("dense_weights", C.c_int32)]


class Profile(C.Structure):
    _fields_ = [("backward_ms", C.c_double), ("backward_launches", C.c_int32),
                ("rollout_ms", C.c_double), ("rollout_launches", C.c_int32),
                ("linearize_ms", C.c_double), ("linearize_launches", C.c_int32),
                ("other_ms", C.c_double), ("other_launches", C.c_int32),
                ("backward_seen", C.c_int32), ("rollout_seen", C.c_int32), ("linearize_seen", C.c_int32),
                ("other_seen", C.c_int32), ("solve_ms", C.c_double), ("solve_launches", C.c_int32), ("solve_seen", C.c_int32)]


_lib = None


def load():
    """Load the HIP library; fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        lib.qilqr_last_error.restype = C.c_char_p
        lib.qilqr_stream.restype = C.c_void_p
        lib.qilqr_host_alloc.restype = C.c_void_p
        lib.qilqr_host_alloc.argtypes = [C.c_size_t]
        lib.qilqr_host_free.argtypes = [C.c_void_p]
        lib.qilqr_sharded_solver.restype = C.c_void_p
        lib.qilqr_sharded_solver.argtypes = [C.c_void_p, C.c_int32]
        lib.qilqr_sharded_destroy.argtypes = [C.c_void_p]
        lib.qilqr_sharded_count.argtypes = [C.c_void_p]
        lib.qilqr_sharded_set_transport.argtypes = [C.c_void_p, C.c_int32]
        lib.qilqr_sharded_transport.argtypes = [C.c_void_p]
        lib.qilqr_sharded_transport.restype = C.c_char_p
        _lib = lib
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def _raise(rc, ls_max_iters=None):
    msg = load().qilqr_last_error().decode()
    if rc == ERR_BAD_INERTIA:
        raise RuntimeError("Inertia matrix is not positive definite!")  # quadrotor_model.cc:23
    if rc == ERR_LENGTH_MISMATCH:
        raise IndexError(msg)  # std::out_of_range from .at(i), cost.hh:39-40
    if rc == ERR_BAD_QUATERNION:
        raise ValueError(msg)
    if rc == ERR_LINE_SEARCH:
        raise RuntimeError(msg)  # ilqr.hh:191-193, same text
    if rc == ERR_INVALID_ARG:
        raise TypeError(msg)
    raise RuntimeError(f"quadrotor_ilqr error {rc}: {msg}")


def _create_args(self, mass_kg, inertia, arm_length_m, torque_to_thrust_ratio_m, g_mpss, Q, R, desired, options, device, profile,
                 sync_every, force_general, single_wave_rollout, precision, streams, persistent, compaction=0, round_launch=0,
                 rounds_per_launch=0, fuse_in_flight=0, dense_weights=0):
    """the C structures of qilqr_create / qilqr_sharded_create; sets self.options and self.desired"""
    m = Model()
    m.mass_kg = mass_kg
    I = _d(inertia)
    if I.shape != (3, 3):
        raise TypeError("inertia must be 3x3")
    for i in range(9):
        m.inertia[i] = I.reshape(9)[i]
    m.arm_length_m = arm_length_m
    m.torque_to_thrust_ratio_m = torque_to_thrust_ratio_m
    m.g_mpss = g_mpss
    Q, R = _d(Q), _d(R)
    if Q.shape != (12, 12) or R.shape != (4, 4):
        raise TypeError("Q must be 12x12 and R 4x4")
    o = Options()
    o.step_update = options["step_update"]
    o.desired_reduction_frac = options["desired_reduction_frac"]
    o.ls_max_iters = int(options["ls_max_iters"])
    o.rtol = options["rtol"]
    o.atol = options["atol"]
    o.max_iters = float(options["max_iters"])
    o.populate_debug = int(bool(options.get("populate_debug", False)))
    self.options = dict(options)
    self.desired = _d(desired).reshape(-1, KNOT)
    dc = DeviceConfig(int(device), int(profile), int(sync_every), int(force_general),
                      int(single_wave_rollout), {"f64": 0, "f32": 1}[precision], int(streams), int(persistent), int(compaction),
                      int(round_launch), int(rounds_per_launch), int(fuse_in_flight), int(dense_weights))
    return m, Q, R, o, dc


def _batch_outputs(init, out):
    """result arrays of a batch solve: fresh, or the caller's (checked)"""
    B = init.shape[0]
    if out is None:
        return dict(traj=np.zeros_like(init), cost=np.zeros(B), **{k: np.zeros(B, dtype=np.int32) for k in ("status", "iters", "n_bwd", "n_fwd")})
    for k, dt, shape in (("traj", np.float64, init.shape), ("cost", np.float64, (B,)), ("status", np.int32, (B,)),
                         ("iters", np.int32, (B,)), ("n_bwd", np.int32, (B,)), ("n_fwd", np.int32, (B,))):
        a = out[k]
        if a.dtype != dt or a.shape != shape or not a.flags["C_CONTIGUOUS"]:
            raise TypeError(f"out[{k!r}] must be a C-contiguous {np.dtype(dt).name} array of shape {shape}")
    return out


class QuadrotorILQRBatch:
    """ILQR<QuadrotorModel> (ilqr.hh:25-41) for batches of independent problems on one MI355X."""

    def __init__(self, mass_kg, inertia, arm_length_m, torque_to_thrust_ratio_m, g_mpss, Q, R, desired,
                 dt_s, options, device=0, profile=0, sync_every=2, force_general=False,
                 single_wave_rollout=False, precision="f64", streams=0, persistent=0, compaction=0, round_launch=0,
                 rounds_per_launch=0, fuse_in_flight=0, dense_weights=0):
        lib = load()
        m, Q, R, o, dc = _create_args(self, mass_kg, inertia, arm_length_m, torque_to_thrust_ratio_m, g_mpss, Q, R, desired, options,
                                      device, profile, sync_every, force_general, single_wave_rollout, precision, streams, persistent, compaction,
                                      round_launch, rounds_per_launch, fuse_in_flight, dense_weights)
        self._h = C.c_void_p()
        rc = lib.qilqr_create_sized(C.byref(m), _p(Q), _p(R), _p(self.desired), C.c_int32(len(self.desired)),
                                    C.c_double(dt_s), C.byref(o), C.byref(dc), C.c_size_t(C.sizeof(dc)), C.byref(self._h))
        if rc:
            self._h = None
            _raise(rc)

    def close(self):
        if getattr(self, "_h", None):
            load().qilqr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- ILQR::solve, one problem (ilqr.hh:53-87) -> (traj, dict(cost, status, iters, debug...))
    def solve(self, init):
        init = _d(init).reshape(-1, KNOT)
        n = len(init)
        out = np.zeros_like(init)
        # the loop of ilqr.hh:58 runs while i < max_iters with max_iters a double: ceil(max_iters) entries at most
        cap = int(min(max(math.ceil(self.options["max_iters"]), 0), 1e6)) if self.options.get("populate_debug") else 0
        dcost = np.zeros(max(cap, 1))
        dtraj = np.zeros((max(cap, 1), n, KNOT))
        cost, st, it, nd = C.c_double(), C.c_int32(), C.c_int32(), C.c_int32()
        rc = load().qilqr_solve(self._h, _p(init), C.c_int32(n), _p(out), C.byref(cost), C.byref(st),
                                C.byref(it), _p(dcost), _p(dtraj), C.c_int32(cap), C.byref(nd))
        if rc:
            _raise(rc)
        k = nd.value
        return out, dict(cost=cost.value, status=st.value, iters=it.value, debug_costs=dcost[:k].copy(),
                         debug_trajs=dtraj[:k].copy())

    # ---- batch of problems, host buffers
    def solve_batch(self, init, desired_batch=None, out=None):
        """qilqr_solve_batch.  `out` (optional): a dict of preallocated result arrays (keys traj, cost, status, iters,
        n_bwd, n_fwd; C-contiguous float64 / int32) that is filled and returned -- with pinned arrays (host_array) the
        copies are direct DMA."""
        init = _d(init)
        B, n = init.shape[0], init.shape[1]
        des = None if desired_batch is None else _d(desired_batch)
        out = _batch_outputs(init, out)
        rc = load().qilqr_solve_batch(self._h, _p(init), _p(des), C.c_int32(B), C.c_int32(n), _p(out["traj"]), _p(out["cost"]),
                                      _ip(out["status"]), _ip(out["iters"]), _ip(out["n_bwd"]), _ip(out["n_fwd"]))
        if rc:
            _raise(rc)
        return out

    # ---- batch of problems, torch CUDA tensors already resident in HBM
    def solve_batch_device(self, init, out_traj, out_cost, out_status, out_iters, out_n_bwd, out_n_fwd,
                           desired_batch=None, wait_current_stream=True):
        """qilqr_solve_batch_device on torch tensors.  Every tensor must live on the solver's device, be
        contiguous and have the ABI's dtype and shape (float64 (B,n,18) / (B,), int32 (B,)); outputs may be None.
        The solve runs on the solver's own stream: it is ordered behind whatever torch has enqueued on its
        current stream (an event recorded here, waited for on the device), and has finished when this returns.
        wait_current_stream=False skips that ordering: only for inputs that are known to be complete (bench.py:
        static inputs, and a gather of the OTHER buffer set still in flight on torch's stream)."""
        import torch
        if init.dim() != 3 or init.shape[2] != KNOT:
            raise TypeError("init must be (B, n, 18)")
        B, n = int(init.shape[0]), int(init.shape[1])
        dev_index = load().qilqr_device(self._h)

        def check(t, name, dtype, shape):
            if t is None:
                return
            if not t.is_cuda or t.device.index != dev_index:
                raise TypeError(f"{name} must be a CUDA tensor on device {dev_index}")
            if t.dtype != dtype:
                raise TypeError(f"{name} must be {dtype}")
            if tuple(t.shape) != shape:
                raise TypeError(f"{name} must have shape {shape}, not {tuple(t.shape)}")
            if not t.is_contiguous():
                raise TypeError(f"{name} must be contiguous")

        check(init, "init", torch.float64, (B, n, KNOT))
        check(desired_batch, "desired_batch", torch.float64, (B, n, KNOT))
        check(out_traj, "out_traj", torch.float64, (B, n, KNOT))
        check(out_cost, "out_cost", torch.float64, (B,))
        for t, name in ((out_status, "out_status"), (out_iters, "out_iters"), (out_n_bwd, "out_n_bwd"),
                        (out_n_fwd, "out_n_fwd")):
            check(t, name, torch.int32, (B,))
        # inputs produced by asynchronous torch work on its current stream: the solver's stream waits for them
        if wait_current_stream:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(init.device))
            rc = load().qilqr_stream_wait_event(self._h, C.c_void_p(ev.cuda_event))
            if rc:
                _raise(rc)
        vp = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
        rc = load().qilqr_solve_batch_device(self._h, vp(init), vp(desired_batch), C.c_int32(B), C.c_int32(n),
                                             vp(out_traj), vp(out_cost), vp(out_status), vp(out_iters),
                                             vp(out_n_bwd), vp(out_n_fwd))
        if rc:
            _raise(rc)

    # ---- the passes the reference tests individually (ilqr_test.cc:102-190), batched
    def cost_trajectory(self, traj):
        traj = _d(traj)
        B, n = traj.shape[0], traj.shape[1]
        cost = np.zeros(B)
        rc = load().qilqr_cost_trajectory(self._h, _p(traj), C.c_int32(B), C.c_int32(n), _p(cost))
        if rc:
            _raise(rc)
        return cost

    def backwards_pass(self, traj):
        traj = _d(traj)
        B, n = traj.shape[0], traj.shape[1]
        gains = np.zeros((B, n, GAIN))
        terms = np.zeros((B, 2))
        rc = load().qilqr_backwards_pass(self._h, _p(traj), C.c_int32(B), C.c_int32(n), _p(gains), _p(terms))
        if rc:
            _raise(rc)
        return gains, terms

    def forward_sim(self, traj, gains, alpha=1.0):
        traj, gains = _d(traj), _d(gains)
        B, n = traj.shape[0], traj.shape[1]
        alpha = _d(np.broadcast_to(alpha, (B,)))
        out = np.zeros_like(traj)
        rc = load().qilqr_forward_sim(self._h, _p(traj), _p(gains), _p(alpha), C.c_int32(B), C.c_int32(n), _p(out))
        if rc:
            _raise(rc)
        return out

    def line_search(self, traj, cost, gains, terms):
        traj, gains, cost, terms = _d(traj), _d(gains), _d(cost), _d(terms)
        B, n = traj.shape[0], traj.shape[1]
        out = np.zeros_like(traj)
        oc, step = np.zeros(B), np.zeros(B)
        st = np.zeros(B, dtype=np.int32)
        rc = load().qilqr_line_search(self._h, _p(traj), _p(cost), _p(gains), _p(terms), C.c_int32(B), C.c_int32(n),
                                      _p(out), _p(oc), _p(step), _ip(st))
        if rc:
            _raise(rc)
        return dict(traj=out, cost=oc, step=step, status=st)

    def cost_history(self, B):
        """(B, cap) array: cost after every completed forward pass of the last batch solve (NaN padded);
        needs options['populate_debug']"""
        cap = C.c_int32()
        load().qilqr_cost_history(self._h, C.c_int32(B), None, C.c_int32(0), C.byref(cap))
        hist = np.zeros((B, max(cap.value, 1)))
        rc = load().qilqr_cost_history(self._h, C.c_int32(B), _p(hist), C.c_int32(hist.shape[1]), C.byref(cap))
        if rc:
            _raise(rc)
        return hist

    # ---- profiling
    def profile_reset(self):
        rc = load().qilqr_profile_reset(self._h)
        if rc:
            _raise(rc)

    def profile_mode(self, mode):
        """0 off, 1 k_backward + k_rollout, 2 every kernel, 3 k_backward only, 4 k_rollout only"""
        rc = load().qilqr_profile_mode(self._h, C.c_int32(int(mode)))
        if rc:
            _raise(rc)

    def describe(self, B):
        """qilqr_describe: in words, the arithmetic and the kernels a batch solve of B problems on this handle uses"""
        buf = C.create_string_buffer(2048)
        rc = load().qilqr_describe(self._h, C.c_int32(int(B)), buf, C.c_size_t(len(buf)))
        if rc:
            _raise(rc)
        return buf.value.decode()

    def compaction_moves(self):
        """trajectories the compaction moved in the last batch solve (qilqr_compaction_moves)"""
        m = C.c_int64()
        rc = load().qilqr_compaction_moves(self._h, C.byref(m))
        if rc:
            _raise(rc)
        return m.value

    def set_regularisation(self, mu_init, mu_factor=10.0, mu_max=1e6):
        """Levenberg-Marquardt restarts (an extension the reference lacks; mu_init = 0 switches it off):
        see qilqr_set_regularisation in include/quadrotor_ilqr.h"""
        rc = load().qilqr_set_regularisation(self._h, C.c_double(mu_init), C.c_double(mu_factor),
                                              C.c_double(mu_max))
        if rc == ERR_INVALID_ARG:
            raise ValueError(load().qilqr_last_error().decode())
        if rc:
            _raise(rc)

    def set_integrator(self, integrator):
        """Runge-Kutta extension (the step sketched at quadrotor_model.cc:51-63; 0 = the reference's explicit Euler)."""
        rc = load().qilqr_set_integrator(self._h, C.c_int32(int(integrator)))
        if rc:
            _raise(rc)

    def profile_get(self):
        p = Profile()
        rc = load().qilqr_profile_get(self._h, C.byref(p))
        if rc:
            _raise(rc)
        return {f: getattr(p, f) for f, _ in Profile._fields_}


class QuadrotorILQRSharded:
    """One batch over several devices from one process (qilqr_sharded_*): contiguous shards in the order of `devices`
    (an ordinal may repeat), one solver, stream and host thread per shard, results in place -- problem by problem identical
    to QuadrotorILQRBatch.solve_batch.  The one-process-per-GPU deployment (torch.distributed) is in sharding.py / bench.py."""

    def __init__(self, mass_kg, inertia, arm_length_m, torque_to_thrust_ratio_m, g_mpss, Q, R, desired, dt_s, options,
                 devices=(0,), profile=0, sync_every=2, force_general=False, single_wave_rollout=False, precision="f64",
                 streams=0, persistent=0, compaction=0, round_launch=0, rounds_per_launch=0, fuse_in_flight=0, dense_weights=0):
        lib = load()
        m, Q, R, o, dc = _create_args(self, mass_kg, inertia, arm_length_m, torque_to_thrust_ratio_m, g_mpss, Q, R, desired, options,
                                      0, profile, sync_every, force_general, single_wave_rollout, precision, streams, persistent, compaction,
                                      round_launch, rounds_per_launch, fuse_in_flight, dense_weights)
        self.devices = [int(d) for d in devices]
        arr = (C.c_int32 * len(self.devices))(*self.devices)
        self._h = C.c_void_p()
        rc = lib.qilqr_sharded_create_sized(C.byref(m), _p(Q), _p(R), _p(self.desired), C.c_int32(len(self.desired)), C.c_double(dt_s),
                                            C.byref(o), C.byref(dc), C.c_size_t(C.sizeof(dc)), arr, C.c_int32(len(self.devices)), C.byref(self._h))
        if rc:
            self._h = None
            _raise(rc)

    def close(self):
        if getattr(self, "_h", None):
            load().qilqr_sharded_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shard_ranges(self, B):
        """[(begin, count)] of every shard for a batch of B (qilqr_shard_range)"""
        out = []
        for r in range(len(self.devices)):
            b0, cnt = C.c_int32(), C.c_int32()
            rc = load().qilqr_shard_range(C.c_int32(B), C.c_int32(len(self.devices)), C.c_int32(r), C.byref(b0), C.byref(cnt))
            if rc:
                _raise(rc)
            out.append((b0.value, cnt.value))
        return out

    def solve_batch(self, init, desired_batch=None, out=None):
        init = _d(init)
        B, n = init.shape[0], init.shape[1]
        des = None if desired_batch is None else _d(desired_batch)
        out = _batch_outputs(init, out)
        rc = load().qilqr_solve_batch_sharded(self._h, _p(init), _p(des), C.c_int32(B), C.c_int32(n), _p(out["traj"]), _p(out["cost"]),
                                              _ip(out["status"]), _ip(out["iters"]), _ip(out["n_bwd"]), _ip(out["n_fwd"]))
        if rc:
            _raise(rc)
        return out


    TRANSPORTS = {"auto": 0, "rccl": 1, "peer_copy": 2}

    def set_transport(self, transport):
        """how qilqr_solve_batch_sharded_device moves the shards' rows to the root: 'auto', 'rccl' (ncclSend / ncclRecv), 'peer_copy'"""
        rc = load().qilqr_sharded_set_transport(self._h, C.c_int32(self.TRANSPORTS[transport]))
        if rc:
            _raise(rc)
        return self.transport()

    def transport(self):
        return load().qilqr_sharded_transport(self._h).decode()

    def solve_batch_gathered(self, init, out_traj, out_cost, out_status, out_iters, out_n_bwd, out_n_fwd, desired_batch=None, root=0):
        """qilqr_solve_batch_sharded_device: host inputs, results gathered into device arrays on the device of shard `root`:
        traj (B, n, 18) float64, cost (B,) float64, the rest (B,) int32; any may be None.  An output is a raw device address
        (int: the caller vouches for its size) or a contiguous torch tensor of that dtype and shape on that device.
        Returns the exposed gather time in ms."""
        init = _d(init)
        B, n = init.shape[0], init.shape[1]
        des = None if desired_batch is None else _d(desired_batch)

        def ptr(t, dtype, shape):
            if t is None:
                return None
            if isinstance(t, int):
                return C.c_void_p(t)
            if str(t.dtype) != dtype or tuple(t.shape) != shape or not t.is_contiguous() or t.device.index != self.devices[root]:
                raise ValueError(f"output tensor must be contiguous {dtype} {shape} on device {self.devices[root]}")
            return C.c_void_p(t.data_ptr())

        ms = C.c_double(0.0)
        rc = load().qilqr_solve_batch_sharded_device(
            self._h, _p(init), _p(des), C.c_int32(B), C.c_int32(n), C.c_int32(root), ptr(out_traj, "torch.float64", (B, n, 18)),
            ptr(out_cost, "torch.float64", (B,)), ptr(out_status, "torch.int32", (B,)), ptr(out_iters, "torch.int32", (B,)),
            ptr(out_n_bwd, "torch.int32", (B,)), ptr(out_n_fwd, "torch.int32", (B,)), C.byref(ms))
        if rc:
            _raise(rc)
        return ms.value


def gather_schedule(B, n, devices, root=0, arrays=63):
    """qilqr_gather_schedule: the transfers of qilqr_solve_batch_sharded_device for these shards, computed without touching a
    device -- list of dicts {shard, array, src_rank, dst_rank, src_off, dst_off, count}"""
    dv = np.ascontiguousarray(devices, dtype=np.int32)
    fn = load().qilqr_gather_schedule
    fn.restype = C.c_int
    args = (C.c_int32(B), C.c_int32(n), dv.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int32(len(dv)), C.c_int32(root), C.c_uint32(arrays))
    cnt = fn(*args, None, C.c_int32(0))
    if cnt < 0:
        raise ValueError("qilqr_gather_schedule: bad arguments")
    out = np.zeros((max(cnt, 1), 7), dtype=np.int64)
    fn(*args, out.ctypes.data_as(C.POINTER(C.c_int64)), C.c_int32(cnt))
    keys = ("shard", "array", "src_rank", "dst_rank", "src_off", "dst_off", "count")
    return [dict(zip(keys, (int(v) for v in row))) for row in out[:cnt]]


def sharded_from_config(cfg, devices=(0,), **kw):
    return QuadrotorILQRSharded(**cfg["model"], Q=cfg["Q"], R=cfg["R"], desired=cfg["desired"], dt_s=cfg["dt"],
                                options=cfg["options"], devices=devices, **kw)


def host_array(shape, dtype=np.float64):
    """A NumPy array in pinned host memory (qilqr_host_alloc): hipMemcpy to / from it is direct DMA.  The memory is
    released when the array (and every view of it) is garbage collected."""
    import weakref
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = load().qilqr_host_alloc(C.c_size_t(max(n, 1)))
    if not p:
        raise MemoryError(load().qilqr_last_error().decode())
    buf = (C.c_char * max(n, 1)).from_address(p)
    a = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    weakref.finalize(buf, load().qilqr_host_free, C.c_void_p(p))
    return a


PIN_ARITHMETIC = dict(single_wave_rollout=3)  # QILQR_PIN_ARITHMETIC of the header: one rollout kernel at every batch size (the backward pass needs no pinning since round 6)


def from_config(cfg, **kw):
    """Build a solver from a quadrotorilqr_amd.problems config dict."""
    m = cfg["model"]
    return QuadrotorILQRBatch(m["mass_kg"], m["inertia"], m["arm_length_m"], m["torque_to_thrust_ratio_m"],
                              m["g_mpss"], cfg["Q"], cfg["R"], cfg["desired"], cfg["dt"], cfg["options"], **kw)
