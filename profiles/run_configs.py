#!/usr/bin/env python3
"""Runs the non-headline BASELINE.json configurations at full size on the GPU box and prints one
JSON line per configuration (wall time of qilqr_solve_batch_device with inputs resident in HBM,
status histogram, pass counts).  Usage: python profiles/run_configs.py [config3] [config3f64] [config4shard] [config5] [config5reg] [big]
(config3 = BASELINE.json configs[2]: B = 8192, N = 200, mixed fp32 / fp64 mode)"""
import json
import os
import sys
import time

# every sub-batch stream gets a hardware queue of its own (set before the HIP runtime starts, as bench.py does: DESIGN.md section 5)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quadrotorilqr_amd import capi, problems as pb  # noqa: E402


def run(name, cfg, reps=2, reg=None, **solver_kw):
    dev = torch.device("cuda", 0)
    B, N = cfg["init"].shape[:2]
    s = capi.from_config(cfg, sync_every=2, **solver_kw)
    if reg:
        s.set_regularisation(*reg)  # Levenberg-Marquardt restarts (extension)
    init = torch.from_numpy(cfg["init"]).to(dev)
    out = torch.empty_like(init)
    cost = torch.empty(B, dtype=torch.float64, device=dev)
    ints = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)]
    # one untimed solve, then `reps` solves back to back (as bench.py times its steps: the clocks stay up between them; a
    # single solve from an idle GPU reads about 10 % lower at these sizes)
    s.solve_batch_device(init, out, cost, *ints)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(max(reps, 1)):
        s.solve_batch_device(init, out, cost, *ints)
    torch.cuda.synchronize()
    best = (time.perf_counter() - t) / max(reps, 1)
    st, it, nb, nf = (x.cpu().numpy() for x in ints)
    s.close()  # (its streams go with it: a later configuration's sub-batch streams then find hardware queues of their own)
    print(json.dumps({"config": name, "B": B, "N": N, **{k: str(v) for k, v in solver_kw.items()}, "seconds": best, "solves_per_s": B / best,
                      "knot_steps_per_s": float((nb.sum() + nf.sum()) * N / best),
                      "status_counts": np.bincount(st, minlength=4).tolist(), "iters_mean": float(it.mean()),
                      "iters_max": int(it.max()), "n_bwd_mean": float(nb.mean()), "n_fwd_mean": float(nf.mean())}))


if __name__ == "__main__":
    which = sys.argv[1:] or ["config4shard", "config5"]
    if len(which) > 1:
        # one process per configuration: HIP hands hardware queues to streams in the order the streams are created and does not
        # give a destroyed stream's place back, so the sub-batch streams of a second solver in the same process can share a
        # queue (B = 8192: 393 000 solves/s behind an earlier solver, 459 000 alone)
        import subprocess
        for name in which:
            subprocess.run([sys.executable, os.path.abspath(__file__), name], check=False)
        sys.exit(0)
    if "config3" in which:  # BASELINE.json configs[2]: fp32 storage / lane-local arithmetic, fp64 recursion and decisions
        run("configs[2] (B=8192, N=200, seed 3), precision f32 (mixed)", pb.config3(), precision="f32")
    if "config3f64" in which:  # the same problems and tolerances in the fp64 mode, for comparison
        run("configs[2] problems (B=8192, N=200, seed 3), precision f64", pb.config3(), precision="f64")
    if "config4shard" in which:  # one rank's shard of configs[3]: 8192 problems, 100 knots, seed 4
        run("configs[3] shard of one GPU (B=8192, N=100, seed 4)", pb.config2(B=8192, N=100, seed=4))
    if "config5" in which:
        a, b = pb.config5()
        run("configs[4] half A: model A hover (B=2048, N=500)", a, reps=1)
        run("configs[4] half B: demo box-climb, random starts (B=2048, N=500)", b, reps=1)
    if "config5reg" in which:  # the same with restarts on, and with one trial per line search (restarts do the damping)
        a, b = pb.config5()
        run("configs[4] half A, restarts (1, x10, <= 1e8) on", a, reps=1, reg=(1.0, 10.0, 1e8))
        a["options"] = dict(a["options"], ls_max_iters=1)
        run("configs[4] half A, ls_max_iters = 1, restarts off", a, reps=1)
        run("configs[4] half A, ls_max_iters = 1, restarts (1, x10, <= 1e8) on", a, reps=1, reg=(1.0, 10.0, 1e8))
        run("configs[4] half B, restarts (1, x10, <= 1e8) on", b, reps=1, reg=(1.0, 10.0, 1e8))
    if "big" in which:
        run("B=65536, N=100, model A (whole configs[3] on one GPU)", pb.config2(B=65536, N=100, seed=4), reps=1)
