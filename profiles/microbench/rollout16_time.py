#!/usr/bin/env python3
"""Diagnostic: average launch time of each rollout kernel (qilqr_device_config.single_wave_rollout = 1 k_rollout,
2 k_rollout3, 3 k_rollout16) through qilqr_forward_sim with every launch timed, for one or more builds of the library.
usage (from the repository root): python profiles/microbench/rollout16_time.py [lib.so ...]"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from quadrotorilqr_amd import capi, problems as pb
libs = sys.argv[1:] or [capi.LIB_PATH]
for lib in libs:
    capi.LIB_PATH = os.path.abspath(lib); capi._lib = None
    for B in (1024, 64, 2048, 4096):
        cfg = pb.config2(B=B, N=100)
        base = capi.from_config(cfg)
        # a rollout as it occurs inside a solve: the third iteration's (the trajectory is already near its nominal one)
        tr = cfg["init"]
        for _ in range(3):
            gains, _ = base.backwards_pass(tr)
            tr_next = base.forward_sim(tr, gains, 1.0)
            tr_prev, tr = tr, tr_next
        gains, _ = base.backwards_pass(tr_prev)
        cfg = dict(cfg, init=tr_prev)
        ref = None
        for kern in (1, 2, 3):
            s = capi.from_config(cfg, profile=2, single_wave_rollout=kern)
            for _ in range(3): got = s.forward_sim(cfg["init"], gains, 1.0)
            s.profile_reset()
            for _ in range(10): s.forward_sim(cfg["init"], gains, 1.0)
            p = s.profile_get()
            if ref is None: ref = got
            print(os.path.basename(lib), "B", B, "kernel", kern, "us/launch", round(1e3 * p["rollout_ms"] / p["rollout_launches"], 2),
                  "max |diff| vs kernel 1: %.2e" % np.abs(got - ref).max(), flush=True)
