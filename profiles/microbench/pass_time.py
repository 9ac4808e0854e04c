#!/usr/bin/env python3
"""Diagnostic: launch times of k_linearize and of the backward kernel with EVERY trajectory live (qilqr_backwards_pass:
linearise + recursion on every trajectory), for builds of the library whose results may be garbage (timing-only variants).
usage (repository root): PYTHONPATH=. python3 profiles/microbench/pass_time.py B name1 name2 ..."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quadrotorilqr_amd import problems as pb  # noqa: E402

B = int(sys.argv[1])
cfg = pb.config2(B=B, N=100, seed=4)
for name in sys.argv[2:]:
    path = os.path.join(ROOT, "quadrotorilqr_amd", "lib", "libquadrotor_ilqr.so" if name == "product" else f"libquadrotor_ilqr_{name}.so")
    spec = importlib.util.spec_from_file_location("capi_" + name, os.path.join(ROOT, "quadrotorilqr_amd", "capi.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.LIB_PATH = path
    fg = {"gfac": 7, "mfac": 8, "fused": 5}.get(name, 4)
    path = path if name not in ("gfac", "mfac", "fused") else os.path.join(ROOT, "quadrotorilqr_amd", "lib", "libquadrotor_ilqr.so")
    m.LIB_PATH = path
    s = m.from_config(cfg, profile=2, force_general=fg)
    for _ in range(2):
        s.backwards_pass(cfg["init"])
    s.profile_reset()
    for _ in range(5):
        s.backwards_pass(cfg["init"])
    p = s.profile_get()
    print(f"{name:10s} B={B}: k_linearize {p['linearize_ms'] * 1e3 / p['linearize_launches']:7.2f} us, k_backward4 {p['backward_ms'] * 1e3 / p['backward_launches']:7.2f} us (every trajectory live)")
    s.close()
