"""Compaction of the live trajectories (qilqr_device_config.compaction; k_compact_plan / k_compact_move): between a round's
backward pass and its rollout the trajectories still running move into a dense prefix of the workspace and the finished ones
they replace leave for the caller's arrays.  A trajectory's arithmetic does not depend on its slot, so every result must be
bit-identical with and without -- at every batch shape, kernel family, precision, with sub-batch streams and with
Levenberg-Marquardt restarts (whose knot records move along) -- and a permutation of the problems must permute the results.
The oracle (tests/test_gpu_parity.py's tolerances) checks the compacted solve itself."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402  (the checker)
from quadrotorilqr_amd import capi, problems as pb  # noqa: E402
from tests.test_gpu_parity import randomised_cfg  # noqa: E402
from tests.observed import observed  # noqa: E402

KEYS = ("traj", "cost", "status", "iters", "n_bwd", "n_fwd")


def solve_device(s, init):
    """a batch solve from pageable host arrays: qilqr_solve_batch stages them and runs the device-resident solve compaction works
    on (no copy-back under the tail without pinned outputs); the finished rows land in the staging arrays as they leave"""
    return s.solve_batch(np.ascontiguousarray(init))


def assert_same(a, b, label=""):
    for k in KEYS:
        np.testing.assert_array_equal(a[k], b[k], err_msg=f"{label}: {k}")


@pytest.mark.parametrize("B,N,kw", [
    (37, 40, {}), (256, 60, {}), (1500, 50, {}), (1500, 50, dict(streams=3)), (700, 30, dict(precision="f32")),
    (600, 40, dict(force_general=4)), (600, 40, dict(force_general=2, single_wave_rollout=1)), (600, 40, dict(single_wave_rollout=2)),
    (300, 40, dict(force_general=1)),
])
def test_compaction_gives_the_same_bits(B, N, kw):
    cfg = pb.config2(B=B, N=N, seed=11)
    on, off = capi.from_config(cfg, compaction=1, **kw), capi.from_config(cfg, compaction=-1, **kw)
    a, b = solve_device(on, cfg["init"]), solve_device(off, cfg["init"])
    assert_same(a, b, f"B={B} {kw}")
    assert (a["status"] >= 0).all() and np.isfinite(a["traj"]).all()  # every row was written
    assert a["iters"].max() > a["iters"].min() + 3  # (the problems do finish in different rounds)
    assert on.compaction_moves() > B // 8 and off.compaction_moves() == 0, (on.compaction_moves(), off.compaction_moves())
    on.close(); off.close()


def test_compacted_solve_matches_the_oracle():
    cfg = pb.config2(B=96, N=40, seed=5)
    s = capi.from_config(cfg, compaction=1)
    out = solve_device(s, cfg["init"])
    o = orc.OracleSolver(orc.model_params(**cfg["model"]), cfg["Q"], cfg["R"], cfg["desired"], cfg["dt"], orc.options(**cfg["options"]))
    ref = o.solve_batch(cfg["init"], n_threads=8)
    for k in ("status", "iters", "n_bwd", "n_fwd"):
        np.testing.assert_array_equal(out[k], ref[k])
    observed("compacted solve, 96 x 40", out, ref)
    np.testing.assert_allclose(out["cost"], ref["cost"], rtol=1e-9)
    np.testing.assert_allclose(out["traj"], ref["traj"], atol=1e-6)


@pytest.mark.parametrize("seed,precision", [(100, "f64"), (101, "f64"), (103, "f64"), (106, "f64"), (101, "f32"), (106, "f32")])
def test_compaction_with_restarts_moves_the_knot_records(seed, precision):
    """Levenberg-Marquardt restarts run the recursion again on the CURRENT records of a trajectory: they move with it (in the mixed
    mode as fp32 pairs)."""
    cfg, reg = randomised_cfg(seed, restarts=True)
    # the randomised batches are small (1..40 problems): tile them so that slots really change
    reps = 8
    init = np.concatenate([cfg["init"]] * reps)
    r = np.random.default_rng(seed)
    init = init[r.permutation(len(init))]
    on, off = capi.from_config(cfg, compaction=1, precision=precision), capi.from_config(cfg, compaction=-1, precision=precision)
    for s in (on, off):
        s.set_regularisation(*reg)
    assert_same(solve_device(on, init), solve_device(off, init), f"seed {seed}")


def test_a_permutation_of_the_problems_permutes_the_results():
    """slot independence, and the rows of the caller's arrays: problem p's result is in row p wherever it was solved"""
    cfg = pb.config2(B=900, N=40, seed=21)
    s = capi.from_config(cfg, compaction=1)
    a = solve_device(s, cfg["init"])
    perm = np.random.default_rng(3).permutation(900)
    b = solve_device(s, cfg["init"][perm])
    for k in KEYS:
        np.testing.assert_array_equal(a[k][perm], b[k], err_msg=k)


def test_the_solver_is_reusable_at_other_batch_sizes():
    cfg = pb.config2(B=300, N=30, seed=8)
    s = capi.from_config(cfg, compaction=1)
    full = solve_device(s, cfg["init"])
    # a smaller and a larger batch on the same handle (workspace regrown: the row table with it)
    small = pb.config2(B=70, N=30, seed=8)
    np.testing.assert_array_equal(solve_device(s, small["init"])["cost"], full["cost"][:70])
    big = pb.config2(B=1100, N=30, seed=8)
    np.testing.assert_array_equal(solve_device(s, big["init"])["cost"][:300], full["cost"])
    np.testing.assert_array_equal(solve_device(s, cfg["init"])["traj"], full["traj"])


@pytest.mark.parametrize("B,N,opts", [(1, 20, {}), (5, 3, {}), (130, 2, {}), (64, 30, dict(max_iters=1)), (200, 25, dict(max_iters=0)),
                                      (300, 30, dict(ls_max_iters=0)), (257, 30, dict(rtol=1e-2, atol=1e-2))])
def test_edge_shapes_and_options(B, N, opts):
    """one problem, a ragged tile, two knots, a single iteration, none at all, no trial allowed (every problem leaves in the same
    round: nothing to move), loose thresholds (most leave at once)"""
    cfg = pb.config2(B=B, N=N, seed=13)
    cfg["options"] = dict(cfg["options"], **opts)
    on, off = capi.from_config(cfg, compaction=1), capi.from_config(cfg, compaction=-1)
    a, b = solve_device(on, cfg["init"]), solve_device(off, cfg["init"])
    assert_same(a, b, f"B={B} N={N} {opts}")
    assert np.isfinite(a["traj"]).all()


def test_full_size_shard_with_and_without():
    """BASELINE.json configs[3], the shard of one GPU (B = 8192, N = 100): the automatic setting compacts; same bits as never."""
    cfg = pb.config2(B=8192, N=100, seed=4)
    auto, never = capi.from_config(cfg), capi.from_config(cfg, compaction=-1)
    assert_same(solve_device(auto, cfg["init"]), solve_device(never, cfg["init"]), "B=8192")
    assert auto.compaction_moves() > 2000 and never.compaction_moves() == 0, auto.compaction_moves()


@pytest.mark.parametrize("B,N,kw", [(4352, 40, {}), (4352, 40, dict(streams=1)), (4608, 40, dict(single_wave_rollout=3)), (4352, 40, dict(precision="f32", single_wave_rollout=3)), (4096, 40, {})])
def test_late_rounds_of_a_large_batch_take_the_combined_launch(B, N, kw):
    """Round 6: a batch beyond 4096 changes over to the combined launch (k_backward_rollout, then k_round with four rounds per launch) once its
    running trajectories fit it AND its rollouts are k_rollout16's by their ordinal (the 17th on; every one with single_wave_rollout = 3) --
    the same bits as three launches per round to the end (round_launch = 1) and as no compaction at all, and fewer launches."""
    cfg = pb.config2(B=B, N=N, seed=12)
    if kw.get("precision") == "f32":
        cfg["options"] = dict(cfg["options"], rtol=1e-5, atol=1e-5)   # (what fp32 storage can resolve: DESIGN.md section 6)
    tail, three, never = (capi.from_config(cfg, profile=1, **kw), capi.from_config(cfg, profile=1, round_launch=1, **kw),
                          capi.from_config(cfg, compaction=-1, **kw))
    a, b, c = solve_device(tail, cfg["init"]), solve_device(three, cfg["init"]), solve_device(never, cfg["init"])
    assert_same(a, b, f"B={B} {kw}: combined launch in the tail / three launches")
    assert_same(a, c, f"B={B} {kw}: / no compaction")
    assert np.isin(a["status"], [0, 1]).all()
    pa, pt = tail.profile_get(), three.profile_get()
    assert pa["rollout_launches"] + 4 <= pt["rollout_launches"], (pa, pt)   # the tail's rounds have no rollout launch of their own
    for h in (tail, three, never):
        h.close()


def test_bad_value_is_refused():
    cfg = pb.config2(B=4, N=10)
    with pytest.raises(TypeError, match="compaction"):
        capi.from_config(cfg, compaction=2)
