#!/usr/bin/env python3
"""Diagnostic (round 6): would configs[1]'s batch of 1024 lose anything if it were solved as k sub-batches, each on k_round launches of its own
(what hiding the upload of the host-buffer call behind the first sub-batch's rounds would need: DESIGN.md section 7a)?  The 1024 problems cut
into k contiguous chunks, one handle (fuse_in_flight = 1: the combined launch although other solves are in flight) and one host thread per
chunk, all started together; wall time until the last chunk is done, median of `reps`.
usage: GPU_MAX_HW_QUEUES=8 PYTHONPATH=. python3 profiles/microbench/parts_feasibility.py [reps]"""
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from quadrotorilqr_amd import capi, problems as pb  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
B = 1024
cfg = pb.config2(B=B, N=100, seed=2)
for k in (1, 2, 4, 1, 2, 4):
    workers = []
    for p in range(k):
        lo, hi = B * p // k, B * (p + 1) // k
        init = torch.from_numpy(cfg["init"][lo:hi]).to(dev)
        s = capi.from_config(cfg, sync_every=2, fuse_in_flight=1)
        out = torch.empty_like(init)
        cost = torch.empty(hi - lo, dtype=torch.float64, device=dev)
        ints = [torch.empty(hi - lo, dtype=torch.int32, device=dev) for _ in range(4)]
        s.solve_batch_device(init, out, cost, *ints)
        workers.append((s, init, out, cost, ints))
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        gate = threading.Barrier(k + 1)

        def run(w):
            s, init, out, cost, ints = w
            gate.wait()
            s.solve_batch_device(init, out, cost, *ints)

        th = [threading.Thread(target=run, args=(w,)) for w in workers]
        for t in th:
            t.start()
        gate.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    rounds = [int(w[4][3].max()) for w in workers]
    print(f"{k} chunk(s): {np.median(ts) * 1e3:7.3f} ms (min {min(ts) * 1e3:.3f})   rounds of the chunks' slowest problems {rounds}", flush=True)
    for w in workers:
        w[0].close()
