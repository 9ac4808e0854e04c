#!/usr/bin/env python3
"""Diagnostic: launch time of the backward kernel with every trajectory live, for several batch sizes, kernel choices and builds.
usage: PYTHONPATH=. python3 profiles/microbench/bw_one.py name:force_general[,name:force_general...] B1 B2 ..."""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quadrotorilqr_amd import problems as pb
mods = {}
for spec in sys.argv[1].split(","):
    name, fg = spec.split(":")
    if name not in mods:
        sp = importlib.util.spec_from_file_location("capi_" + name, os.path.join(ROOT, "quadrotorilqr_amd", "capi.py"))
        m = importlib.util.module_from_spec(sp); sp.loader.exec_module(m)
        m.LIB_PATH = os.path.join(ROOT, "quadrotorilqr_amd", "lib", "libquadrotor_ilqr.so" if name == "product" else f"libquadrotor_ilqr_{name}.so")
        mods[name] = m
    for B in map(int, sys.argv[2:]):
        cfg = pb.config2(B=B, N=100)
        s = mods[name].from_config(cfg, force_general=int(fg), profile=3)
        tr = s.forward_sim(cfg["init"], np.zeros((B, 100, 52)), 1.0)
        for _ in range(3): s.backwards_pass(tr)
        s.profile_reset()
        for _ in range(20): s.backwards_pass(tr)
        p = s.profile_get()
        print(f"{name:10s} force_general={fg} B={B:5d}: {p['backward_ms'] * 1e3 / p['backward_launches']:7.2f} us")
        s.close()
