#!/usr/bin/env python3
"""Diagnostic: does k_rollout16 run slower while k_linearize runs beside it on another stream?  (The question behind a
linearisation that chases the rollout instead of following it.)  Needs the diagnostic build:
  make -C quadrotorilqr_amd/csrc variant NAME=diag DEFS=-DQILQR_DIAG
usage (from the repository root): python profiles/microbench/beside.py [B ...]"""
import ctypes as C, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
sys.path.insert(0, os.getcwd())
from quadrotorilqr_amd import capi, problems as pb
capi.LIB_PATH = os.path.abspath("quadrotorilqr_amd/lib/libquadrotor_ilqr_diag.so"); capi._lib = None
for B in [int(x) for x in sys.argv[1:]] or [1024, 256, 2048]:
    cfg = pb.config2(B=B, N=100)
    s = capi.from_config(cfg)
    tr = cfg["init"]
    for _ in range(3):
        gains, _ = s.backwards_pass(tr)
        tr_prev, tr = tr, s.forward_sim(tr, gains, 1.0)
    gains, _ = s.backwards_pass(tr_prev)
    s.forward_sim(tr_prev, gains, 1.0)  # leaves the state the diagnostic entry re-runs
    f = capi.load().qilqr_debug_rollout_beside_linearize
    us = (C.c_float * 2)()
    for beside in (0, 1, 0, 1):
        for _ in range(2):
            rc = f(s._h, C.c_int32(B), C.c_int32(100), C.c_int32(20), C.c_int32(beside), us)
        assert rc == 0, rc
        print(f"B {B}  rollout {'beside k_linearize' if beside else 'alone':18s} {us[0]:7.2f} us per launch" + (f"   (k_linearize {us[1]:.2f} us per launch)" if beside else ""), flush=True)
    s.close()
