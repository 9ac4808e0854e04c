#!/usr/bin/env python3
"""Diagnosis (round 4) of round 3's anomaly: with the kernel bodies as __device__ functions -- global
accesses then become flat ones, the argument structures go through scratch -- the untouched six-wavefront k_backward4 left the
oracle's path on 3 of 60 restart seeds.  Two builds side by side in one process: the product (bodies included into the kernel
functions) and -DQILQR_BODY_AS_FUNCTION; the randomised restart problems of tests/test_gpu_parity.py, force_general = 4.
For every seed on which a build's counts differ from the oracle's: the comparison of the oracle's path that explains it
(tests/exit_paths.py) with its margin; and whether ONE backward pass of the two builds on the same trajectory gives the same bits.
(The build switch -DQILQR_BODY_AS_FUNCTION -- k_backward4's body as a __device__ function, QILQR_BODY_REF_MASK choosing which argument
structures it takes by reference -- was in csrc/ilqr_kernels.h until the cause was found and fixed at the source (commit "Root cause of the
flat-pointer anomaly ..."): it is gone with the reason for it; this script documents how the nine deviating seeds were found.)
usage (repository root, GPU box; the variant: make -C quadrotorilqr_amd/csrc variant NAME=bodyfn DEFS=-DQILQR_BODY_AS_FUNCTION):
    PYTHONPATH=. python3 profiles/microbench/flat_anomaly.py [first_seed [n_seeds]]"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests import test_gpu_parity as T  # noqa: E402
from tests.exit_paths import explain, describe  # noqa: E402


def binding(name):
    spec = importlib.util.spec_from_file_location("capi_" + name, os.path.join(ROOT, "quadrotorilqr_amd", "capi.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.LIB_PATH = os.path.join(ROOT, "quadrotorilqr_amd", "lib", "libquadrotor_ilqr.so" if name == "product" else f"libquadrotor_ilqr_{name}.so")
    return m


first = int(sys.argv[1]) if len(sys.argv) > 1 else 12
count = int(sys.argv[2]) if len(sys.argv) > 2 else 120
libs = {n: binding(n) for n in ("product", "bodyfn")}
keys = ("status", "iters", "n_bwd", "n_fwd")
differ_pass, differ_solve, off_oracle = 0, 0, {n: 0 for n in libs}
for seed in range(first, first + count):
    cfg, reg = T.randomised_cfg(seed, restarts=True)
    o = T.oracle_for(cfg)
    o.set_regularisation(*reg)
    ref = o.solve_batch(cfg["init"], n_threads=8)
    outs, passes = {}, {}
    for n, m in libs.items():
        s = m.from_config(cfg, force_general=4)
        s.set_regularisation(*reg)
        passes[n] = s.backwards_pass(cfg["init"])
        outs[n] = s.solve_batch(cfg["init"])
        s.close()
    same_pass = all(np.array_equal(a, b) for a, b in zip(passes["product"], passes["bodyfn"]))
    same_solve = all(np.array_equal(outs["product"][k], outs["bodyfn"][k]) for k in keys + ("cost", "traj"))
    differ_pass += not same_pass
    differ_solve += not same_solve
    for n in libs:
        bad = np.zeros(len(ref["status"]), dtype=bool)
        for k in keys:
            bad |= outs[n][k] != ref[k]
        if bad.any():
            off_oracle[n] += 1
            for b in np.nonzero(bad)[0]:
                got = tuple(int(outs[n][k][b]) for k in keys)
                r = o.solve_decisions(cfg["init"][b])
                try:
                    d = explain(got, r)
                    print(f"seed {seed} {n}: " + describe(int(b), got, r, d))
                except AssertionError as e:
                    print(f"seed {seed} {n}: problem {b} UNEXPLAINED: {str(e)[:400]}")
    if not same_pass or not same_solve:
        g = [float(np.max(np.abs(a - b)) / max(np.max(np.abs(a)), 1e-300)) for a, b in zip(passes["product"], passes["bodyfn"])]
        print(f"seed {seed}: one backward pass of the two builds: gains differ by {g[0]:.2e} of the largest, terms by {g[1]:.2e}; "
              f"whole solves the same bits: {same_solve}; max rel cost difference {float(np.max(np.abs(outs['product']['cost'] - outs['bodyfn']['cost']) / np.abs(outs['product']['cost']))):.2e}")
print(f"{count} seeds from {first}: one backward pass differs between the builds on {differ_pass}, whole solves on {differ_solve}; "
      f"seeds with counts off the oracle's: {off_oracle}")
