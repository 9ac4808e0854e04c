#!/usr/bin/env python3
"""Diagnostic: throughput with several independent batch solves in flight on one GPU (one solver handle and one
host thread per batch in flight; ctypes releases the GIL during the call).  A stream of 1024-problem batches is
the serving case: the tail of one batch (few trajectories still iterating) overlaps the head of the next.
usage: inflight.py [batch [steps [fuse_in_flight]]]"""
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from quadrotorilqr_amd import capi, problems as pb  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
FUSE = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # 1: keep the combined launch (k_round) although other solves are in flight
dev = torch.device("cuda", 0)
cfg = pb.config2(B=B, N=100, seed=2)
init = torch.from_numpy(cfg["init"]).to(dev)
for inflight in (1, 2, 3, 4):
    workers = []
    for _ in range(inflight):
        s = capi.from_config(cfg, sync_every=2, fuse_in_flight=FUSE)
        out = torch.empty_like(init)
        cost = torch.empty(B, dtype=torch.float64, device=dev)
        ints = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)]
        s.solve_batch_device(init, out, cost, *ints)  # warm-up
        workers.append((s, out, cost, ints))
    torch.cuda.synchronize()
    per = steps // inflight

    def run(w):
        s, out, cost, ints = w
        for _ in range(per):
            s.solve_batch_device(init, out, cost, *ints)

    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(w,)) for w in workers]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"B {B}  {inflight} in flight: {per * inflight * B / dt:10.0f} solves/s  ({dt * 1e3 / (per * inflight):.3f} ms per batch)")
    for w in workers:
        w[0].close()
