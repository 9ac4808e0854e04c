"""The Runge-Kutta extension on the device (qilqr_set_integrator(1)) against the oracle's statement of it (orc_set_integrator(1)):
every pass, and whole solves -- same bars as the Euler path's parity tests."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402  (the checker)
from quadrotorilqr_amd import capi, problems as pb  # noqa: E402
from tests.test_gpu_parity import assert_same_exit_paths, oracle_for, random_cfg  # noqa: E402


@pytest.mark.parametrize("seed,dense", [(31, False), (32, True), (33, "qsym"), (34, "sym")])
def test_rk4_passes_match_oracle(seed, dense):
    cfg = random_cfg(seed, n=25, dense=dense, B=6)
    trajs = cfg["init"]
    s, o = capi.from_config(cfg), oracle_for(cfg)
    s.set_integrator(1)
    o.set_integrator(1)
    B = len(trajs)
    np.testing.assert_allclose(s.cost_trajectory(trajs), [o.cost_trajectory(t) for t in trajs], rtol=1e-12)
    gains, terms = s.backwards_pass(trajs)
    r = np.random.default_rng(seed)
    alpha = 0.5 ** r.integers(0, 3, B)
    fwd = s.forward_sim(trajs, gains, alpha)
    for b in range(B):
        g, t = o.backwards_pass(trajs[b])
        scale = np.abs(g).max()
        np.testing.assert_allclose(gains[b], g, rtol=1e-8, atol=1e-9 * scale)
        np.testing.assert_allclose(terms[b], t, rtol=1e-8, atol=1e-10 * max(1.0, np.abs(t).max()))
        np.testing.assert_allclose(fwd[b], o.forward_sim(trajs[b], gains[b], alpha[b]), rtol=0, atol=1e-9)
    # the Euler results come back when the extension is switched off again
    e = capi.from_config(cfg)
    s.set_integrator(0)
    g0, t0 = s.backwards_pass(trajs)
    g1, t1 = e.backwards_pass(trajs)
    np.testing.assert_array_equal(g0, g1)
    np.testing.assert_array_equal(t0, t1)
    assert np.abs(g0 - gains).max() > 1e-6


@pytest.mark.parametrize("seed", [0, 1, 2, 5])
def test_rk4_solves_match_oracle(seed):
    from tests import test_gpu_parity as tp
    r = np.random.default_rng(4000 + seed)
    cfg = pb.config2(B=int(r.integers(3, 40)), N=int(r.integers(10, 70)), seed=50 + seed)
    if seed % 2:
        G = r.uniform(-1, 1, (12, 12))
        cfg["Q"] = cfg["Q"] + 0.3 * (G @ G.T)
    cfg["options"] = dict(cfg["options"], rtol=1e-10, atol=1e-10)
    s, o = capi.from_config(cfg), oracle_for(cfg)
    s.set_integrator(1)
    o.set_integrator(1)
    if seed == 5:  # Levenberg-Marquardt restarts on top
        cfg["options"]["ls_max_iters"] = 2
        s, o = capi.from_config(cfg), oracle_for(cfg)
        s.set_integrator(1)
        o.set_integrator(1)
        s.set_regularisation(1.0, 4.0, 1e6)
        o.set_regularisation(1.0, 4.0, 1e6)
    out = s.solve_batch(cfg["init"])
    ref = o.solve_batch(cfg["init"], n_threads=8)
    assert_same_exit_paths(out, ref, o, cfg["init"])
    same = out["iters"] == ref["iters"]
    np.testing.assert_allclose(out["cost"][same], ref["cost"][same], rtol=1e-8)
    np.testing.assert_allclose(out["traj"][same], ref["traj"][same], atol=1e-6)
    # and the integrator matters: the Euler solve of the same problems ends elsewhere
    eul = capi.from_config(cfg).solve_batch(cfg["init"])
    assert np.abs(eul["traj"] - out["traj"]).max() > 1e-4


def test_rk4_refuses_mixed_precision_and_bad_values():
    cfg = pb.config2(B=2, N=8)
    with pytest.raises(TypeError, match="precision 0"):
        capi.from_config(cfg, precision="f32").set_integrator(1)
    with pytest.raises(TypeError, match="integrator must be"):
        capi.from_config(cfg).set_integrator(2)
    # persistent = 1 with the extension quietly takes the rounds (k_solve4 is the Euler path; diagnostics build)
    from tests.diag_lib import capi_diag
    s = capi_diag().from_config(cfg, persistent=1, profile=1)
    s.set_integrator(1)
    s.solve_batch(cfg["init"])
    assert s.profile_get()["solve_launches"] == 0
