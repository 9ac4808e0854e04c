#!/usr/bin/env python3
"""Randomised parity soak: the randomised GPU test of tests/test_gpu_parity.py over many more seeds, with every
backward kernel of the product (automatic, k_backward4 six-wavefront and fused, one wavefront, general -- with Eigen's pivoted
LDL^T since round 3) and rollout kernel forced in turn, sub-batches on three streams, the compaction of the live trajectories forced (round 4), and with Levenberg-Marquardt restarts on
(few trials per line search).  (k_backward2 and k_solve4 live in the diagnostics build: tests/test_gpu_parity.py covers them.)
usage: python profiles/microbench/soak.py [first_seed [n_seeds]]"""
import sys

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import test_gpu_parity as T  # noqa: E402
from quadrotorilqr_amd import capi  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 12
count = int(sys.argv[2]) if len(sys.argv) > 2 else 120
orig = capi.from_config
bad = 0
for label, kw in [("automatic", {}), ("k_backward4 six waves", dict(force_general=4)), ("six waves, G factors", dict(force_general=7)), ("six waves, M factors", dict(force_general=8)),
                  ("k_backward4 fused", dict(force_general=5)),
                  ("one wavefront", dict(force_general=2)), ("general", dict(force_general=1)),
                  ("k_rollout", dict(single_wave_rollout=1)), ("k_rollout3", dict(single_wave_rollout=2)), ("k_rollout16", dict(single_wave_rollout=3)),
                  ("three streams", dict(streams=3)), ("compaction", dict(compaction=1)), ("compaction, six waves, k_rollout3", dict(compaction=1, force_general=4, single_wave_rollout=2)),
                  ("compaction, three streams", dict(compaction=1, streams=3)), ("restarts, compaction", dict(compaction=1)),
                  ("restarts, compaction, six waves", dict(compaction=1, force_general=4)), ("restarts", dict()), ("restarts, six waves", dict(force_general=4)), ("restarts, six waves, G factors", dict(force_general=7)), ("restarts, fused", dict(force_general=5)),
                  ("restarts, one wavefront", dict(force_general=2)), ("restarts, general", dict(force_general=1))]:
    capi.from_config = lambda cfg, _kw=kw, **k: orig(cfg, **{**_kw, **k})
    fails = 0
    for seed in range(first, first + count):
        try:
            T.test_randomised_models_and_horizons_match_oracle(seed, restarts=label.startswith("restarts"))
        except AssertionError as e:
            fails += 1
            print(label, "seed", seed, "FAILED:", str(e).splitlines()[:4])
    print(f"{label:34s}: {count - fails}/{count} seeds agree with the oracle")
    bad += fails
capi.from_config = orig
print("soak done, failures:", bad)
