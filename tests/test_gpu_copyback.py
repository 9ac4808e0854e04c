"""qilqr_solve_batch with pinned result arrays copies the finished trajectories back to the host while the tail rounds of
the slowest problems still run (EarlyOut in csrc/ilqr_capi.hip), and puts the late finishers into their rows afterwards -- since round 6
k_gather writes them straight into the caller's mapped arrays over the link; the staged form (compact block, one copy, the host's scatter)
is the fallback for arrays that do not map and is exercised through the diagnostics build.
The caller's arrays must be what the one-piece copy gives, bit for bit, in every case: the early path taken, not taken
(every problem ends in the same round), pageable outputs (never taken), absent outputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from quadrotorilqr_amd import capi, problems as pb  # noqa: E402

KEYS = ("status", "iters", "n_bwd", "n_fwd", "cost", "traj")


def pinned_out(B, n):
    out = dict(traj=capi.host_array((B, n, 18)), cost=capi.host_array((B,)),
               **{k: capi.host_array((B,), np.int32) for k in ("status", "iters", "n_bwd", "n_fwd")})
    out["traj"][...] = np.nan
    out["cost"][...] = np.nan
    for k in ("status", "iters", "n_bwd", "n_fwd"):
        out[k][...] = -7
    return out


@pytest.mark.parametrize("B,n,per_problem_desired", [(1024, 100, False), (777, 60, True), (2048, 50, False)])
def test_two_part_copy_back_is_the_one_piece_copy(B, n, per_problem_desired):
    cfg = pb.config2(B=B, N=n, seed=2)
    desired = None
    if per_problem_desired:
        r = np.random.default_rng(5)
        desired = np.repeat(cfg["desired"][None], B, axis=0)
        desired[:, :, 1:4] += r.uniform(-0.2, 0.2, (B, 1, 3))
    s = capi.from_config(cfg)
    ref = s.solve_batch(cfg["init"].copy(), desired)           # pageable outputs: the one-piece copy
    assert ref["iters"].max() > ref["iters"].mean() + 2      # a tail: the early part has something to overlap
    for rep in range(3):                                       # buffers are kept between calls
        out = pinned_out(B, n)
        got = s.solve_batch(cfg["init"], desired, out=out)
        assert got["traj"] is out["traj"]
        for k in KEYS:
            np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    # the device-resident entry point agrees (it never takes this path)
    import ctypes as C
    lib = capi.load()
    hin = capi.host_array(cfg["init"].shape)
    hin[...] = cfg["init"]
    out = pinned_out(B, n)
    s.solve_batch(hin, desired, out=out)                       # pinned inputs too
    for k in KEYS:
        np.testing.assert_array_equal(out[k], ref[k], err_msg=k)
    assert lib.qilqr_abi_version() >= 5 and C.sizeof(C.c_double) == 8


def test_staged_late_part_is_the_same_copy():
    """the fallback of the late part (arrays that are pinned but do not map into the device's address space), forced in the diagnostics build"""
    from tests.diag_lib import capi_diag
    d = capi_diag()
    cfg = pb.config2(B=1024, N=100, seed=2)
    s = d.from_config(cfg)
    ref = s.solve_batch(cfg["init"].copy())
    try:
        assert d.load().qilqr_debug_set_staged_late(1) == 0
        got = s.solve_batch(cfg["init"], out={k: d.host_array(v.shape, v.dtype) for k, v in pinned_out(1024, 100).items()})
    finally:
        d.load().qilqr_debug_set_staged_late(0)
    direct = s.solve_batch(cfg["init"], out={k: d.host_array(v.shape, v.dtype) for k, v in pinned_out(1024, 100).items()})
    for k in KEYS:
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
        np.testing.assert_array_equal(direct[k], ref[k], err_msg=k)


def test_copy_back_when_every_problem_ends_in_the_same_round():
    """max_iters = 3: every trajectory leaves in the same round, the count of running trajectories falls from B to 0 and the
    early part never starts; and a batch too small for it"""
    cfg = pb.config2(B=512, N=80, seed=7)
    cfg["options"] = dict(cfg["options"], max_iters=3.0)
    s = capi.from_config(cfg)
    ref = s.solve_batch(cfg["init"].copy())
    assert (ref["status"] == 2).all()
    got = s.solve_batch(cfg["init"], out=pinned_out(512, 80))
    for k in KEYS:
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    small = pb.config2(B=40, N=30, seed=7)
    s2 = capi.from_config(small)
    ref2 = s2.solve_batch(small["init"].copy())
    got2 = s2.solve_batch(small["init"], out=pinned_out(40, 30))
    for k in KEYS:
        np.testing.assert_array_equal(got2[k], ref2[k], err_msg=k)


def test_a_bad_quaternion_is_still_refused_before_anything_is_solved():
    cfg = pb.config2(B=300, N=60, seed=3)
    s = capi.from_config(cfg)
    bad = cfg["init"].copy()
    bad[123, 7, 4:8] *= 1.01
    out = pinned_out(300, 60)
    with pytest.raises(ValueError, match="quaternion"):
        s.solve_batch(bad, out=out)
    assert np.isnan(out["traj"]).all() and (out["status"] == -7).all()   # outputs untouched
    good = s.solve_batch(cfg["init"], out=out)                           # and the handle still works
    ref = capi.from_config(cfg).solve_batch(cfg["init"].copy())
    for k in KEYS:
        np.testing.assert_array_equal(good[k], ref[k], err_msg=k)
