#!/usr/bin/env python3
"""Diagnostic (round 6): what a ROUND of a lone trajectory costs, by problem -- BASELINE.json configs[0] (the demo: model D,
box-climb with the roll stepping to pi, initial = desired) against configs[1]'s problems solved one at a time (model A hover, random starts)
and the demo at shorter horizons; B = 1 through qilqr_solve_batch (no debug ring), microseconds per round = wall time / rollouts.
usage (repository root): PYTHONPATH=. python3 profiles/microbench/single_round.py"""
import time
import numpy as np
from quadrotorilqr_amd import capi, problems as pb


def run(name, cfg, init):
    s = capi.from_config(cfg)
    out = s.solve_batch(init)
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        out = s.solve_batch(init)
        best = min(best, time.perf_counter() - t)
    rounds = int(out["n_fwd"][0]) + 1
    print(f"{name:58s} {best * 1e3:7.2f} ms  iters {int(out['iters'][0]):3d}  rollouts {int(out['n_fwd'][0]):3d}  status {int(out['status'][0])}  "
          f"{best * 1e6 / rounds:6.1f} us per round")
    s.close()


c0 = pb.config1(10.0)
run("configs[0]: demo, 100 knots", c0, c0["init"])
c40 = pb.config1(4.0)
run("demo, 40 knots", c40, c40["init"])
c1 = pb.config2(B=1024, N=100, seed=2)
for b in (0, 1, 2):
    run(f"configs[1] problem {b} alone (model A hover, 100 knots)", c1, c1["init"][b:b + 1])
# the demo's model and desired trajectory with a small roll only (no pose error near pi): the same horizon, Log / Exp on their series
d = pb.config1(10.0)
small = d["desired"].copy()
ang = 2.0 * np.arctan2(small[:, 5], small[:, 4]) * 0.2   # a fifth of the roll
small[:, 4], small[:, 5] = np.cos(ang / 2), np.sin(ang / 2)
d2 = dict(d, desired=small, init=small[None].copy())
run("demo with a fifth of the roll (100 knots)", d2, d2["init"])
