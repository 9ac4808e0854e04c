#!/usr/bin/env python3
"""Diagnostic: solve time of each of the eight 1024-problem shards that bench.py --gpus 8 rotates over the ranks (weak
scaling): the N = 1 line solves shard 0 only, so value_8 / (8 value_1) carries the ratio t(shard 0) / mean t(shard).
usage (from the repository root): python profiles/microbench/shard_times.py"""
import sys, time
import torch
sys.path.insert(0, ".")
from quadrotorilqr_amd import capi, problems as pb
dev = torch.device("cuda", 0)
B, N = 1024, 100
ts = []
for sh in range(8):
    cfg = pb.config2(B=B, N=N, seed=2, b0=sh * B)
    init = torch.from_numpy(cfg["init"]).to(dev)
    out = torch.empty_like(init); cost = torch.empty(B, dtype=torch.float64, device=dev)
    ints = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)]
    s = capi.from_config(cfg)
    for _ in range(3):
        s.solve_batch_device(init, out, cost, *ints)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        s.solve_batch_device(init, out, cost, *ints)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 10
    ts.append(t)
    it = ints[1].cpu().numpy(); nf = ints[3].cpu().numpy()
    print(f"shard {sh}: {t * 1e3:.3f} ms  ({B / t:.0f} solves/s)  iterations max {it.max()} mean {it.mean():.1f}  rollouts max {nf.max()}", flush=True)
    s.close()
print(f"shard 0 / mean: {ts[0] / (sum(ts) / 8):.3f}")
