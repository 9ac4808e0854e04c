"""CPU-side checks of the C ABI: the library loads, exports every symbol the header declares,
and the host-only logic (argument validation, error text) behaves like the reference's
constructor.  No compute call is made here (that needs a GPU: tests/test_gpu_*.py)."""
import os
import re

import numpy as np
import pytest

from quadrotorilqr_amd import capi, problems as pb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__ as g
        g.build()


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "quadrotor_ilqr.h")).read()
    declared = set(re.findall(r"\b(qilqr_[a-z_]+)\s*\(", header))
    assert declared == set(capi.EXPORTS)
    lib = capi.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.qilqr_abi_version() == 7


def test_struct_layouts_match_header():
    import ctypes as C
    assert C.sizeof(capi.Model) == 13 * 8
    assert C.sizeof(capi.Options) == 56      # 2 doubles, int32 + pad, 3 doubles, int32 + pad
    assert capi.Options.rtol.offset == 24 and capi.Options.populate_debug.offset == 48
    assert C.sizeof(capi.DeviceConfig) == 52  # thirteen int32 (ABI version 2 added `streams`, 3 `persistent`, 6 `compaction`, 7 the four A/B switches)
    assert C.sizeof(capi.Profile) == 96  # four (double, int32 + pad) pairs, four int32 counts, (double, int32, int32)


def test_bad_inertia_raises_runtime_error_with_reference_text():
    cfg = pb.config2(B=1, N=4)
    bad = dict(cfg["model"], inertia=np.diag([1.0, -1.0, 1.0]))
    with pytest.raises(RuntimeError, match="Inertia matrix is not positive definite!"):
        capi.from_config(dict(cfg, model=bad))
    asym = np.eye(3)
    asym[0, 1] = 0.2
    with pytest.raises(RuntimeError, match="Inertia matrix is not positive definite!"):
        capi.from_config(dict(cfg, model=dict(cfg["model"], inertia=asym)))


def test_unnormalised_desired_quaternion_is_value_error():
    cfg = pb.config2(B=1, N=4)
    d = cfg["desired"].copy()
    d[2, 4] = 1.1
    with pytest.raises(ValueError, match="quaternion"):
        capi.from_config(dict(cfg, desired=d))


def test_wrong_shapes_are_type_errors():
    cfg = pb.config2(B=1, N=4)
    with pytest.raises(TypeError):
        capi.from_config(dict(cfg, Q=np.eye(6)))
    with pytest.raises(TypeError):
        capi.from_config(dict(cfg, model=dict(cfg["model"], inertia=np.eye(2))))


def test_no_cpu_fallback():
    """Without a GPU the product refuses to construct a solver; it never computes on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = pb.config2(B=1, N=4)
    with pytest.raises(RuntimeError, match="no HIP device|no CPU path"):
        capi.from_config(cfg)


def test_sharded_handle_refuses_without_a_device_and_names_the_shard():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = pb.config2(B=1, N=4)
    with pytest.raises(RuntimeError, match=r"shard 0 \(device 0\).*(no HIP device|no CPU path)"):
        capi.sharded_from_config(cfg, devices=[0, 1])
    with pytest.raises(TypeError, match="bad device list"):
        capi.sharded_from_config(cfg, devices=[])


def test_c_abi_shard_ranges_are_the_ones_of_the_multi_process_deployment():
    """qilqr_shard_range (the in-process sharded solve) against sharding.shard_range (one process per GPU): the same
    contiguous, ragged, unpadded partition of a batch -- host arithmetic, no device needed."""
    import ctypes as C
    from quadrotorilqr_amd import sharding
    lib = capi.load()
    for B in (0, 1, 3, 7, 8, 1001, 1024, 65536):
        for world in (1, 2, 3, 6, 8):
            for r in range(world):
                b0, cnt = C.c_int32(), C.c_int32()
                assert lib.qilqr_shard_range(C.c_int32(B), C.c_int32(world), C.c_int32(r), C.byref(b0), C.byref(cnt)) == 0
                lo, hi = sharding.shard_range(B, r, world)
                assert (b0.value, cnt.value) == (lo, hi - lo)
    b0, cnt = C.c_int32(), C.c_int32()
    assert lib.qilqr_shard_range(C.c_int32(8), C.c_int32(2), C.c_int32(2), C.byref(b0), C.byref(cnt)) != 0  # shard out of range
    assert lib.qilqr_shard_range(C.c_int32(8), C.c_int32(0), C.c_int32(0), C.byref(b0), C.byref(cnt)) != 0


def test_product_does_not_reference_the_oracle():
    """The product path (package, include/, src/) may not import, link or call oracle/."""
    bad = []
    for base in ("quadrotorilqr_amd", "include", "src"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hip", ".cpp", ".cc", "Makefile")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"ilqr_oracle|from oracle|import oracle|orc_", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_gather_schedule_of_eight_distinct_devices_against_a_mock_table():
    """The send / receive schedule of qilqr_solve_batch_sharded_device for EIGHT distinct devices (no such node in this pool:
    the multi-rank RCCL path has not run on hardware) against a table worked out here from the documented rule -- contiguous
    shards, the first B % shards one problem larger; a shard's device = its communicator rank; rows go from element 0 of the
    shard's staging buffers to the shard's rows of the root's arrays; the four int32 arrays sit [4][count] in one staging
    block -- for ragged batches, a root in the middle, repeated ordinals (two shards on one device share a rank), fewer
    problems than shards, and a subset of the outputs.  No device is touched (qilqr_gather_schedule computes, nothing else)."""
    def mock(B, n, devices, root, arrays):
        uniq = []
        for d in devices:
            if d not in uniq:
                uniq.append(d)
        rank = [uniq.index(d) for d in devices]
        k = len(devices)
        rows = []
        for r in range(k):
            cnt = B // k + (1 if r < B % k else 0)
            b0 = r * (B // k) + min(r, B % k)
            if cnt == 0:
                continue
            if arrays & 1:
                rows.append(dict(shard=r, array=0, src_rank=rank[r], dst_rank=rank[root], src_off=0, dst_off=b0 * n * 18, count=cnt * n * 18))
            if arrays & 2:
                rows.append(dict(shard=r, array=1, src_rank=rank[r], dst_rank=rank[root], src_off=0, dst_off=b0, count=cnt))
            for q in range(4):
                if arrays & (4 << q):
                    rows.append(dict(shard=r, array=2 + q, src_rank=rank[r], dst_rank=rank[root], src_off=q * cnt, dst_off=b0, count=cnt))
        return rows

    cases = [(65536, 100, list(range(8)), 0, 63), (65537, 100, list(range(8)), 3, 63), (1000, 37, [7, 6, 5, 4, 3, 2, 1, 0], 7, 63),
             (5, 10, list(range(8)), 2, 63), (4099, 50, [0, 1, 0, 1, 2, 2, 3, 3], 4, 63), (8192, 100, list(range(8)), 0, 1 | 2),
             (777, 20, [0, 1, 2, 3, 4, 5, 6, 7], 5, 4 | 32)]
    for B, n, devices, root, arrays in cases:
        got = capi.gather_schedule(B, n, devices, root, arrays)
        assert got == mock(B, n, devices, root, arrays), (B, n, devices, root, arrays)
        # the pieces of one array tile the root's array exactly once, in shard order, and every rank is a device of the list
        for a in range(6):
            pcs = [p for p in got if p["array"] == a]
            if not pcs:
                continue
            unit = n * 18 if a == 0 else 1
            assert pcs[0]["dst_off"] == 0 and sum(p["count"] for p in pcs) == B * unit
            for p, q in zip(pcs, pcs[1:]):
                assert q["dst_off"] == p["dst_off"] + p["count"] and q["shard"] > p["shard"]
        assert all(0 <= p["src_rank"] < len(set(devices)) and p["dst_rank"] == got[0]["dst_rank"] for p in got)
    # the shard rule is the one-process-per-GPU rule (sharding.shard_range)
    from quadrotorilqr_amd import sharding
    for B, k in ((65537, 8), (5, 8), (1000, 3)):
        sched = capi.gather_schedule(B, 1, list(range(k)), 0, 2)
        assert [(p["dst_off"], p["dst_off"] + p["count"]) for p in sched] == [sharding.shard_range(B, r, k) for r in range(k) if sharding.shard_range(B, r, k)[1] > sharding.shard_range(B, r, k)[0]]
    with pytest.raises(ValueError):
        capi.gather_schedule(10, 5, [0, 1], root=2)
