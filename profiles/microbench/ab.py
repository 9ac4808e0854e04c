#!/usr/bin/env python3
"""Diagnostic: A/B of builds of the library on ONE box, in ONE process, interleaved (boxes of the pool differ by a few
percent; runs minutes apart on one box by about one): whole device-resident batch solves (no events: the number `value`
is made of) and, in a second pass, the average launch time of each kernel of the rounds (events on every launch).
usage (repository root): PYTHONPATH=. python3 profiles/microbench/ab.py [B=1024] [N=100] [seed=2] name1=path1.so name2=path2.so ...
       (a bare name means quadrotorilqr_amd/lib/libquadrotor_ilqr_<name>.so; "product" the product library)"""
import importlib.util
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quadrotorilqr_amd import problems as pb  # noqa: E402


def binding(path, tag):
    spec = importlib.util.spec_from_file_location("capi_" + tag, os.path.join(ROOT, "quadrotorilqr_amd", "capi.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.LIB_PATH = path
    return m


def main():
    B, N, seed, reps, extra = 1024, 100, 2, 5, {}
    libs = []
    for a in sys.argv[1:]:
        k, _, v = a.partition("=")
        if k == "B":
            B = int(v)
        elif k == "N":
            N = int(v)
        elif k == "seed":
            seed = int(v)
        elif k == "reps":
            reps = int(v)
        elif k in ("force_general", "single_wave_rollout", "streams", "persistent"):
            extra[k] = int(v)
        else:
            path = v or (os.path.join(ROOT, "quadrotorilqr_amd", "lib", "libquadrotor_ilqr.so") if k == "product" else
                         os.path.join(ROOT, "quadrotorilqr_amd", "lib", f"libquadrotor_ilqr_{k}.so"))
            libs.append((k, path))
    cfg = pb.config2(B=B, N=N, seed=seed)
    dev = torch.device("cuda", 0)
    init = torch.from_numpy(cfg["init"]).to(dev)
    bufs = (torch.empty_like(init), torch.empty(B, dtype=torch.float64, device=dev), [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)])
    solvers = {}
    for name, path in libs:
        m = binding(path, name)
        solvers[name] = (m.from_config(cfg, device=0, **extra), m.from_config(cfg, device=0, profile=2, **extra))
    ref_cost = None
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:  # clocks out of idle
        for name, _ in libs:
            solvers[name][0].solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
    times = {name: [] for name, _ in libs}
    for _ in range(reps):
        for name, _ in libs:
            s = solvers[name][0]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                s.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
            torch.cuda.synchronize()
            times[name].append((time.perf_counter() - t0) / 4)
            c = bufs[1].cpu().numpy()
            if ref_cost is None:
                ref_cost = c
            times.setdefault("_dc", {})[name] = float(np.max(np.abs(c - ref_cost) / np.abs(ref_cost)))
    for name, _ in libs:
        sp = solvers[name][1]
        for _ in range(2):
            sp.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
        sp.profile_reset()
        for _ in range(4):
            sp.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
        p = sp.profile_get()
        t = np.array(times[name]) * 1e3
        print(f"{name:12s} B={B} N={N}: {np.median(t):7.3f} ms per solve (min {t.min():.3f}, max {t.max():.3f}) = {B / np.median(t) * 1e3:9.0f} solves/s | "
              f"backward {p['backward_ms'] * 1e3 / max(p['backward_launches'], 1):6.2f} us  rollout {p['rollout_ms'] * 1e3 / max(p['rollout_launches'], 1):6.2f} us  "
              f"linearize {p['linearize_ms'] * 1e3 / max(p['linearize_launches'], 1):6.2f} us  ({p['backward_launches'] // 4} rounds) | max rel cost diff vs first {times['_dc'][name]:.1e}")


if __name__ == "__main__":
    main()
