"""The reference's Python surface end to end on the GPU: what quadrotor_ilqr.py:256-312 does
(build messages, construct QuadrotorILQR positionally, solve(desired), read debug costs), and the
reference's only end-to-end test, quadrotor_ilqr_test.py:8 (the demo runs without raising)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_golden.npz"))


def test_demo_through_the_binding_matches_golden():
    from src.demo import main
    out = main(4.0)
    np.testing.assert_allclose(out["costs"], G["demo40_cost_hist"], rtol=1e-8)
    np.testing.assert_allclose(out["optimized"], G["demo40_traj"], atol=1e-6)
    assert len(out["iters"]) == len(out["costs"]) == int(G["demo40_meta"][1])
    np.testing.assert_array_equal(out["iters"][-1], out["optimized"])  # solve returns the last accepted rollout


def test_config1_demo_at_its_own_100_knots_through_the_binding():
    """BASELINE.json configs[0] as it is specified -- the demo with horizon_s = 10.0, i.e. 100 knots, default ILQROptions
    (quadrotor_ilqr.py:257-306 with the horizon of SURVEY.md section 8b) -- through the reference's Python surface: messages
    in, QuadrotorILQR(...) positional, solve(desired), messages out.  The problem is chaotic in the reference algorithm itself
    (tests/test_oracle_golden.py::test_demo100_is_chaotic: the desired roll sits on the Log branch cut), so it is held to the
    bar of tests/test_gpu_parity.py::test_demo100_config1: the same exit path and iteration count as the oracle, one
    ILQRIterDebug per iteration, the cost history within 1e-3."""
    from src.demo import main
    out = main(10.0)
    status, iters = (int(v) for v in G["demo100_meta"][:2])  # max_iters, 100
    assert status == 2 and len(out["costs"]) == len(out["iters"]) == iters == 100
    np.testing.assert_allclose(out["costs"], G["demo100_cost_hist"], rtol=1e-3)
    assert out["costs"][-1] < out["costs"][0] / 50
    assert out["optimized"].shape == (100, 18) and out["desired"].shape == (100, 18)
    np.testing.assert_array_equal(out["optimized"][:, 0], out["desired"][:, 0])          # time_s passes through (ilqr.hh:164)
    np.testing.assert_array_equal(out["optimized"][0, 1:14], out["desired"][0, 1:14])    # knot 0 state is the input's (:156)
    np.testing.assert_array_equal(out["iters"][-1], out["optimized"])                    # the last accepted rollout is returned


def test_binding_types_and_debug_switch():
    import src.ilqr_debug_pb2 as dbg
    import src.trajectory_pb2 as traj
    from quadrotorilqr_amd import problems as pb
    from src.demo import options_message, trajectory_message
    from src.quadrotor_ilqr_binding import QuadrotorILQR
    cfg = pb.config2(B=1, N=25)
    m = cfg["model"]
    des = trajectory_message(cfg["desired"])
    o = options_message(dict(cfg["options"], populate_debug=False))
    ilqr = QuadrotorILQR(m["mass_kg"], m["inertia"], m["arm_length_m"], m["torque_to_thrust_ratio_m"], m["g_mpss"],
                         cfg["Q"], cfg["R"], des, cfg["dt"], o)
    t, d = ilqr.solve(trajectory_message(cfg["init"][0]))
    assert isinstance(t, traj.QuadrotorTrajectory) and isinstance(d, dbg.QuadrotorILQRDebug)
    assert len(t.points) == 25 and len(d.iter_debugs) == 0           # empty unless populate_debug
    assert t.points[3].time_s == pytest.approx(0.3)
    t2, _ = ilqr.solve(trajectory_message(cfg["init"][0]))           # the handle is re-usable
    assert t2 == t
    longer = trajectory_message(np.concatenate([cfg["init"][0], cfg["init"][0][-1:]]))
    with pytest.raises(IndexError):
        ilqr.solve(longer)                                           # cost.hh:39-40


def test_line_search_exhaustion_raises_reference_text():
    from quadrotorilqr_amd import problems as pb
    from src.demo import options_message, trajectory_message
    from src.quadrotor_ilqr_binding import QuadrotorILQR
    cfg = pb.config2(B=1, N=20)
    m = cfg["model"]
    o = options_message(dict(cfg["options"], desired_reduction_frac=10.0, ls_max_iters=7))
    ilqr = QuadrotorILQR(m["mass_kg"], m["inertia"], m["arm_length_m"], m["torque_to_thrust_ratio_m"], m["g_mpss"],
                         cfg["Q"], cfg["R"], trajectory_message(cfg["desired"]), cfg["dt"], o)
    with pytest.raises(RuntimeError, match=r"^Reached maximum number of line search iterations, 7\n$"):
        ilqr.solve(trajectory_message(cfg["init"][0]))


def test_legacy_create_symbol_reads_only_the_fields_every_header_had():
    """qilqr_device_config grew (ABI 6: compaction; ABI 7: round_launch ...) and qilqr_create copied the whole
    current structure, so a caller built against an older, shorter header had the library read past its structure.  Since ABI
    version 7 the exported symbol qilqr_create reads the 32 bytes every earlier header had, and qilqr_create_sized takes the
    caller's size: a 32-byte structure followed by garbage is accepted by both when the size says 32, and the garbage is
    rejected by name when the size says it is part of the structure."""
    import ctypes as C
    from quadrotorilqr_amd import capi, problems as pb
    cfg = pb.config2(B=4, N=8)
    lib = capi.load()
    holder = capi.QuadrotorILQRBatch.__new__(capi.QuadrotorILQRBatch)
    m, Q, R, o, dc = capi._create_args(holder, **cfg["model"], Q=cfg["Q"], R=cfg["R"], desired=cfg["desired"], options=cfg["options"],
                                       device=0, profile=0, sync_every=2, force_general=0, single_wave_rollout=0, precision="f64",
                                       streams=0, persistent=0, compaction=77, round_launch=99)  # fields 9 and 10: garbage
    args = (C.byref(m), capi._p(Q), capi._p(R), capi._p(holder.desired), C.c_int32(len(holder.desired)), C.c_double(cfg["dt"]), C.byref(o), C.byref(dc))
    for call in (lambda h: lib.qilqr_create(*args, C.byref(h)), lambda h: lib.qilqr_create_sized(*args, C.c_size_t(32), C.byref(h))):
        h = C.c_void_p()
        assert call(h) == 0, lib.qilqr_last_error()
        lib.qilqr_destroy(h)
    h = C.c_void_p()
    assert lib.qilqr_create_sized(*args, C.c_size_t(C.sizeof(dc)), C.byref(h)) == capi.ERR_INVALID_ARG
    assert b"round_launch" in lib.qilqr_last_error() or b"compaction" in lib.qilqr_last_error()  # (the garbage is seen, and named)
    assert lib.qilqr_create_sized(*args, C.c_size_t(30), C.byref(h)) == capi.ERR_INVALID_ARG  # not a whole number of fields


def test_describe_says_which_arithmetic_a_handle_uses():
    """the arithmetic a caller gets is chosen by whether Q and R are bit-exactly symmetric; qilqr_describe says which"""
    from quadrotorilqr_amd import capi, problems as pb
    cfg = pb.config2(B=4, N=8)
    d = capi.from_config(cfg).describe(1024)
    assert "symmetric-weight forms" in d and "k_backward4, fused" in d and "k_rollout16" in d and "k_round" in d and "4 rounds per launch" in d
    d = capi.from_config(cfg).describe(8192)
    assert "six wavefronts" in d and "k_rollout3" in d and "three launches" in d and "sub-batch streams: " in d and "compaction of the running trajectories: on" in d
    assert "three launches, then k_round (the same bits" in d and "block(s) of four per CU and the rollouts are k_rollout16's (round 16 on)" in d
    assert "factored by the gradient wavefront" in d and "first 16 rollouts, k_rollout16 from there on" in d
    # a mid-size batch: the fused kernels, the compaction first and the combined launch once the running trajectories fit (the text
    # follows run_solve's predicate -- no "one launch" beside "compaction: on"), and describing is READ-ONLY: the answers do not depend on
    # what was asked before, and a solve between two questions changes nothing
    h = capi.from_config(cfg)
    d2048, d64 = h.describe(2048), h.describe(64)
    assert "k_backward4, fused" in d2048 and "three launches while the compaction runs, then k_round" in d2048 and "fit 1 block(s) of four per CU" in d2048 and "one launch (k_round)" not in d2048
    assert "one launch (k_round)" in d64 and "compaction: off" in d64
    h.solve_batch(cfg["init"])
    assert h.describe(2048) == d2048 and h.describe(64) == d64 and h.describe(8192) == d
    d = capi.from_config(cfg, force_general=1).describe(1024)
    assert "the reference's own forms" in d and "general kernel" in d and "three launches" in d
    Q = cfg["Q"].copy()
    Q[0, 1] += 1e-3  # not symmetric any more: the general kernel without being asked
    assert "the reference's own forms" in capi.from_config(dict(cfg, Q=Q)).describe(16)
    assert "mixed precision" in capi.from_config(cfg, precision="f32").describe(16)
