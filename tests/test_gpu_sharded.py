"""qilqr_solve_batch_sharded: one batch cut into contiguous shards, one solver / stream / host thread per shard, in one
process.  On the one-GPU test box every shard goes to device 0 (an ordinal may repeat): the code path -- ragged shard
ranges, slices of the caller's arrays, concurrent host threads, error propagation -- is the one an 8-GPU node takes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from quadrotorilqr_amd import capi, problems as pb  # noqa: E402


@pytest.mark.parametrize("B,shards", [(203, 3), (5, 8), (64, 2), (1, 1)])
def test_sharded_solve_is_the_single_device_solve_problem_by_problem(B, shards):
    cfg = pb.config2(B=B, N=40, seed=21)
    r = np.random.default_rng(B)
    desired_batch = np.repeat(cfg["desired"][None], B, axis=0)
    desired_batch[:, :, 1:4] += r.uniform(-0.2, 0.2, (B, 1, 3))
    one = capi.from_config(cfg)
    many = capi.sharded_from_config(cfg, devices=[0] * shards)
    ranges = many.shard_ranges(B)
    assert ranges[0][0] == 0 and sum(c for _, c in ranges) == B and all(a[0] + a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    for desired in (None, desired_batch):
        a = one.solve_batch(cfg["init"], desired)
        b = many.solve_batch(cfg["init"], desired)
        for k in ("status", "iters", "n_bwd", "n_fwd", "cost", "traj"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    # results in place in pinned arrays of the caller
    out = dict(traj=capi.host_array(cfg["init"].shape), cost=capi.host_array((B,)),
               **{k: capi.host_array((B,), np.int32) for k in ("status", "iters", "n_bwd", "n_fwd")})
    c = many.solve_batch(cfg["init"], None, out=out)
    a = one.solve_batch(cfg["init"])
    assert c["traj"] is out["traj"]
    np.testing.assert_array_equal(c["traj"], a["traj"])
    np.testing.assert_array_equal(c["iters"], a["iters"])


def test_sharded_create_by_mask_and_accessors():
    import ctypes as C
    cfg = pb.config2(B=9, N=12, seed=3)
    lib = capi.load()
    probe = capi.sharded_from_config(cfg, devices=[0, 0])
    assert lib.qilqr_sharded_count(probe._h) == 2
    s1 = lib.qilqr_sharded_solver(probe._h, 1)
    assert s1 and lib.qilqr_device(C.c_void_p(s1)) == 0 and not lib.qilqr_sharded_solver(probe._h, 2)
    # the mask form: bit 0 -> one shard on device 0; an empty mask and an absent device are errors with a text
    m, Q, R, o, dc = capi._create_args(probe, **cfg["model"], Q=cfg["Q"], R=cfg["R"], desired=cfg["desired"], options=cfg["options"],
                                       device=0, profile=0, sync_every=2, force_general=False, single_wave_rollout=0,
                                       precision="f64", streams=0, persistent=0)
    h = C.c_void_p()
    args = (C.byref(m), capi._p(Q), capi._p(R), capi._p(probe.desired), C.c_int32(len(probe.desired)), C.c_double(cfg["dt"]),
            C.byref(o), C.byref(dc))
    assert lib.qilqr_sharded_create_mask(*args, C.c_uint64(1), C.byref(h)) == 0
    assert lib.qilqr_sharded_count(h) == 1
    lib.qilqr_sharded_destroy(h)
    assert lib.qilqr_sharded_create_mask(*args, C.c_uint64(0), C.byref(h)) != 0
    assert b"empty device mask" in lib.qilqr_last_error()
    assert lib.qilqr_sharded_create_mask(*args, C.c_uint64(1 << 63), C.byref(h)) != 0
    assert b"shard 0 (device 63)" in lib.qilqr_last_error()
    # the mask form carries the caller's structure size too (qilqr_sharded_create_mask_sized; in C source the macro of the plain
    # name).  compaction = -1 sits behind the 32 bytes of ABI version 5: honoured through the sized call, ignored -- as the header says --
    # through the raw symbol that binaries built before version 7 bind
    m, Q, R, o, dc = capi._create_args(probe, **cfg["model"], Q=cfg["Q"], R=cfg["R"], desired=cfg["desired"], options=cfg["options"],
                                       device=0, profile=0, sync_every=2, force_general=False, single_wave_rollout=0,
                                       precision="f64", streams=0, persistent=0, compaction=-1)
    args = (C.byref(m), capi._p(Q), capi._p(R), capi._p(probe.desired), C.c_int32(len(probe.desired)), C.c_double(cfg["dt"]),
            C.byref(o), C.byref(dc))
    texts = {}
    for name, call in (("sized", lambda: lib.qilqr_sharded_create_mask_sized(*args, C.c_size_t(C.sizeof(dc)), C.c_uint64(1), C.byref(h))),
                       ("raw", lambda: lib.qilqr_sharded_create_mask(*args, C.c_uint64(1), C.byref(h)))):
        assert call() == 0, lib.qilqr_last_error()
        buf = C.create_string_buffer(2048)
        assert lib.qilqr_describe(C.c_void_p(lib.qilqr_sharded_solver(h, 0)), C.c_int32(2048), buf, C.c_size_t(len(buf))) == 0
        texts[name] = buf.value.decode()
        lib.qilqr_sharded_destroy(h)
    assert "compaction: off" in texts["sized"], texts["sized"]
    assert "compaction of the running trajectories: on" in texts["raw"], texts["raw"]
    assert lib.qilqr_sharded_create_mask_sized(*args, C.c_size_t(7), C.c_uint64(1), C.byref(h)) != 0  # not a structure size


def test_a_failing_shard_reports_itself_and_the_others_complete():
    B = 12
    cfg = pb.config2(B=B, N=10, seed=4)
    many = capi.sharded_from_config(cfg, devices=[0, 0, 0])
    bad = cfg["init"].copy()
    bad[9, 3, 4:8] *= 1.5  # problem 9 is in shard 2 (problems 8..11): not a unit quaternion
    out = dict(traj=np.zeros_like(bad), cost=np.zeros(B), **{k: np.full(B, -1, dtype=np.int32) for k in ("status", "iters", "n_bwd", "n_fwd")})
    with pytest.raises(ValueError, match=r"shard 2 \(device 0\)"):
        many.solve_batch(bad, None, out=out)
    assert (out["status"][:8] >= 0).all() and (out["status"][8:] == -1).all()  # shards 0 and 1 solved, shard 2 untouched
    good = many.solve_batch(cfg["init"])
    np.testing.assert_array_equal(good["status"][:8], out["status"][:8])


class _Hip:
    """device buffers without torch: the HIP runtime this process already runs on, through ctypes (loading PyTorch's second
    ROCm runtime into a process that has already initialised /opt/rocm's is what these tests must not depend on)"""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.lib = C.CDLL("libamdhip64.so.7")
        self.ptrs = []

    def alloc(self, nbytes, fill=0xFF):
        p = self.C.c_void_p()
        assert self.lib.hipMalloc(self.C.byref(p), self.C.c_size_t(nbytes)) == 0
        assert self.lib.hipMemset(p, self.C.c_int(fill), self.C.c_size_t(nbytes)) == 0
        assert self.lib.hipDeviceSynchronize() == 0
        self.ptrs.append(p)
        return p.value

    def download(self, ptr, shape, dtype):
        a = np.empty(shape, dtype=dtype)
        assert self.lib.hipMemcpy(a.ctypes.data_as(self.C.c_void_p), self.C.c_void_p(ptr), self.C.c_size_t(a.nbytes), self.C.c_int(2)) == 0
        return a

    def close(self):
        for p in self.ptrs:
            self.lib.hipFree(p)
        self.ptrs = []


@pytest.mark.parametrize("transport", ["rccl", "peer_copy", "auto"])
@pytest.mark.parametrize("B,shards,root", [(203, 3, 0), (5, 8, 2), (64, 2, 1)])
def test_results_gathered_into_one_devices_memory(transport, B, shards, root):
    """qilqr_solve_batch_sharded_device: every shard's rows go from its solver's staging buffers straight to their place in
    the root device's arrays -- ragged shards, more shards than problems -- over RCCL (ncclSend / ncclRecv inside one group;
    on this one-GPU box every shard sits on device 0, so the communicator has ONE rank and every transfer is a send and a
    receive on it: the calls, the grouping, the stream ordering behind each shard's `done` event and the offsets are those
    of an 8-GPU node, the wire is not) or by peer copies; bit-identical to the single-device solve either way."""
    cfg = pb.config2(B=B, N=40, seed=21)
    one = capi.from_config(cfg)
    a = one.solve_batch(cfg["init"])
    many = capi.sharded_from_config(cfg, devices=[0] * shards)
    said = many.set_transport(transport)
    if transport == "rccl":
        assert said.startswith("rccl: ncclSend / ncclRecv, 1 rank")
    else:
        assert said.startswith("peer copies")   # 'auto' with every shard on one device: nothing to communicate
    hip = _Hip()
    shapes = dict(traj=((B, 40, 18), np.float64), cost=((B,), np.float64),
                  **{k: ((B,), np.int32) for k in ("status", "iters", "n_bwd", "n_fwd")})
    out = {k: hip.alloc(int(np.prod(sh)) * np.dtype(dt).itemsize) for k, (sh, dt) in shapes.items()}
    get = lambda k: hip.download(out[k], *shapes[k])
    try:
        for rep in range(2):  # the second call reuses communicators, streams and events
            ms = many.solve_batch_gathered(cfg["init"], out["traj"], out["cost"], out["status"], out["iters"], out["n_bwd"], out["n_fwd"],
                                           root=root)
            assert 0.0 <= ms < 1e3
            for k in ("status", "iters", "n_bwd", "n_fwd", "cost", "traj"):
                np.testing.assert_array_equal(get(k), a[k], err_msg=k)
        # any output may be absent; a bad root is refused
        many.solve_batch_gathered(cfg["init"], None, out["cost"], None, None, None, None, root=root)
        np.testing.assert_array_equal(get("cost"), a["cost"])
        with pytest.raises(Exception):
            many.solve_batch_gathered(cfg["init"], None, out["cost"], None, None, None, None, root=shards)
        # a failing shard names itself, nothing stays in flight, and the handle still works afterwards
        bad = cfg["init"].copy()
        bad[B - 1, 3, 4:8] *= 1.5
        with pytest.raises(Exception) as ei:
            many.solve_batch_gathered(bad, out["traj"], out["cost"], out["status"], out["iters"], out["n_bwd"], out["n_fwd"], root=root)
        assert "shard" in str(ei.value) and "quaternion" in str(ei.value)
        many.solve_batch_gathered(cfg["init"], out["traj"], None, None, None, None, None, root=root)
        np.testing.assert_array_equal(get("traj"), a["traj"])
    finally:
        many.close()
        hip.close()


def test_plain_c_host_gathers_over_rccl():
    """tests/c/sharded_gather_harness.c: a C program with include/quadrotor_ilqr.h, the HIP runtime and nothing else in the
    process solves a batch sharded three ways, gathers it into the root's memory over RCCL and over peer copies, compares
    with the single-device solve bit for bit and reports the exposed gather time."""
    import json
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    exe = os.path.join(here, "c", "sharded_gather_harness")
    src = exe + ".c"
    hdr = os.path.join(root, "include", "quadrotor_ilqr.h")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["gcc", "-O2", "-std=c99", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(root, "include"),
                               src, "-L" + os.path.join(root, "quadrotorilqr_amd", "lib"), "-lquadrotor_ilqr", "-L/opt/rocm/lib",
                               "-lamdhip64", "-lm", "-Wl,-rpath," + os.path.join(root, "quadrotorilqr_amd", "lib"),
                               "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    for transport, said in ((1, "rccl: ncclSend / ncclRecv"), (2, "peer copies")):
        # (1536 problems: shards of 512 take the same backward kernel as the whole batch -- which kernel a batch size selects
        # shows in the last bits of a trajectory, tests/test_gpu_parity.py::test_config2_full_size_properties)
        r = subprocess.run([exe, str(transport), "1536", "100", "1", "0", "0", "0"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert j["bit_identical"] is True and j["transport"].startswith(said) and j["shards"] == 3 and j["converged"] == 1536
        assert 0 <= j["gather_ms_best_of_3"] < 100
        print(j)


def test_pinned_arithmetic_gives_batching_independent_bits():
    """A problem's bits and the size of its batch.  Since round 6 every form of k_backward4 performs one arithmetic,
    so the BACKWARD PASS NEEDS NO PINNING: with nothing but the rollout kernel fixed -- QILQR_PIN_ARITHMETIC (include/quadrotor_ilqr.h;
    capi.PIN_ARITHMETIC) = k_rollout16 at every size, or k_rollout3 at every size -- and the backward kernel left to the automatic choice
    (fused up to 4096 trajectories per call, six wavefronts beyond and in the early rounds of a mid-size batch, Q_uu factored by the gradient
    or by the matrix wavefronts as the running count goes), a problem's result is the same bits whether it is solved in a batch of 4500
    (beyond the 4096 at which the automatic choice changes kernels), in slices of 1500, 300 or alone, through sub-batch streams, or
    sharded.  What the automatic ROLLOUT choice does not promise is stated at the end."""
    cfg = pb.config2(B=4500, N=24, seed=12)
    keys = ("traj", "cost", "status", "iters", "n_bwd", "n_fwd")
    for pin in (capi.PIN_ARITHMETIC, dict(single_wave_rollout=2)):
        whole = capi.from_config(cfg, **pin).solve_batch(cfg["init"])
        assert np.isin(whole["status"], [0, 1]).all()
        for lo, hi in ((0, 1500), (1500, 3000), (3000, 4500), (4100, 4400), (17, 18), (4499, 4500)):
            part = capi.from_config(cfg, **pin).solve_batch(cfg["init"][lo:hi])
            for k in keys:
                np.testing.assert_array_equal(part[k], whole[k][lo:hi], err_msg=f"{pin} {k} [{lo}:{hi}]")
        streams = capi.from_config(cfg, streams=3, **pin).solve_batch(cfg["init"])
        sh = capi.sharded_from_config(cfg, devices=(0, 0, 0), **pin).solve_batch(cfg["init"])
        for k in keys:
            np.testing.assert_array_equal(streams[k], whole[k], err_msg=k)
            np.testing.assert_array_equal(sh[k], whole[k], err_msg=k)
    # the automatic choice on ONE side of 4096: the rollout kernel goes by the rollout's ordinal (k_rollout3 for the first 16 rollouts of a
    # trajectory, k_rollout16 from the 17th on), which is the round in every call -- the same bits in a batch of 4500, of 4200, on three
    # sub-batch streams
    cfg_long = pb.config2(B=4500, N=40, seed=12)
    a = capi.from_config(cfg_long).solve_batch(cfg_long["init"])
    assert a["n_fwd"].max() > 20 and np.isin(a["status"], [0, 1]).all()   # some problems roll out on both kernels
    b = capi.from_config(cfg_long).solve_batch(cfg_long["init"][:4200])
    c = capi.from_config(cfg_long, streams=3).solve_batch(cfg_long["init"])
    for k in keys:
        np.testing.assert_array_equal(b[k], a[k][:4200], err_msg=k)
        np.testing.assert_array_equal(c[k], a[k], err_msg=k)
    # the automatic choice across the 4096 boundary changes the ROLLOUT kernels of a trajectory's first 16 rollouts (k_rollout16 /
    # k_rollout3): not bit-identical in general, the same results to rounding
    auto_whole = capi.from_config(cfg).solve_batch(cfg["init"])
    auto_part = capi.from_config(cfg).solve_batch(cfg["init"][:1500])
    np.testing.assert_allclose(auto_part["cost"], auto_whole["cost"][:1500], rtol=1e-9)
