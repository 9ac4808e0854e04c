#!/usr/bin/env python3
"""bench.py -- iLQR solves/sec on BASELINE.json configs[1]: B=1024 random SE(3) starts per GPU,
100 knots, fp64, model A.

One "step" = one batched solve of the whole per-GPU batch, inputs already resident in HBM.
N>1: launched by torch.distributed.run, one rank per GPU; each rank solves its own shard of
1024 problems (weak scaling, no data-path collective) and the converged trajectories are
gathered on rank 0 over RCCL inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     for the dominant kernel, from HIP events recorded on the solver's stream inside
               the timed region (algorithmic flops/bytes from BASELINE.md section 4)
  cpu_baseline the CPU oracle timed on this host's cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector = matrix peak (AMD CDNA4 datasheet)
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8 TB/s
# algorithmic work per knot
FLOP_BWD_KNOT, FLOP_FWD_KNOT = 30000.0, 1250.0
BYTES_BWD_KNOT, BYTES_FWD_KNOT, BYTES_IO_KNOT = 86 * 8.0, 103 * 8.0, 53 * 8.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="problems per GPU")
    ap.add_argument("--knots", type=int, default=100)
    ap.add_argument("--sync-every", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-all", action="store_true", help="HIP events around every kernel, not only the two candidates for dominant kernel")
    ap.add_argument("--rollout", type=int, default=-1, help="rollout kernel: 0 pose + control + loader waves, 1 single wave (default: library default)")
    ap.add_argument("--backward", type=int, default=0, help="diagnostic: backward kernel (qilqr_device_config.force_general: 0 automatic, 1 general, 2 one wavefront per trajectory)")
    ap.add_argument("--streams", type=int, default=0, help="sub-batches on their own streams (qilqr_device_config.streams; 0 automatic)")
    ap.add_argument("--event-stride", type=int, default=4, help="time every k-th launch of the dominant kernel in the timed region (a timed dispatch costs the stream about 6 us)")
    ap.add_argument("--settle-ms", type=float, default=300.0, help="untimed solves before the warm-up steps (clocks out of idle)")
    ap.add_argument("--no-serving", action="store_true", help="skip the extra several-batches-in-flight measurement (never part of value)")
    ap.add_argument("--serving-batches", type=int, default=18)
    ap.add_argument("--no-profile", action="store_true", help="diagnostic: no HIP events around the kernels (roofline = null)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from quadrotorilqr_amd import capi, problems as pb, sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # test hook (tests/test_gpu_bench_contract.py): run the N > 1 code path with every rank on GPU 0 and gloo,
    # because a one-GPU box cannot host two RCCL ranks; never set by the driver
    one_device_test = os.environ.get("QILQR_BENCH_ONE_DEVICE_TEST") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0 if one_device_test else local_rank)
        dist.init_process_group("gloo" if one_device_test else "nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if (world > 1 and not one_device_test) else 0)
    to_wire = (lambda t: t.cpu()) if one_device_test else (lambda t: t)  # gloo gathers host tensors

    B, N = args.batch, args.knots
    cfg = pb.config2(B=B, N=N, seed=2, b0=rank * B)  # counter-based generator: shard-independent
    solver = capi.from_config(cfg, device=dev.index, profile=(0 if args.no_profile else (2 if args.profile_all else 1)), sync_every=args.sync_every,
                              force_general=args.backward, streams=args.streams, **({} if args.rollout < 0 else dict(single_wave_rollout=args.rollout)))

    init = torch.from_numpy(cfg["init"]).to(dev)
    # N > 1: the global batch of a step is the N shards of 1024 distinct problems (shard k = problems
    # k*B .. (k+1)*B of the counter-based generator); rank r solves shard (r + step) mod N.  How long a shard
    # takes is set by its slowest problem (31 to 45 rollouts over the first eight shards), so a fixed
    # assignment would make every step wait for the same unlucky rank; rotating it evens the ranks' totals
    # over the steps without any exchange (sharding.shard_of_step).
    inits = {rank: init}
    for sh in range(world):
        if sh not in inits:
            inits[sh] = torch.from_numpy(pb.config2(B=B, N=N, seed=2, b0=sh * B)["init"]).to(dev)
    # two sets of output buffers, used alternately: with N > 1 the gather of step s (RCCL, torch's stream) runs
    # while the solver's own stream is already solving step s + 1
    out_traj = [torch.empty_like(init) for _ in range(2)]
    out_cost = [torch.empty(B, dtype=torch.float64, device=dev) for _ in range(2)]
    out_i = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)]  # status, iters, n_bwd, n_fwd
    sizes = [B] * world
    step_no = [0]

    gathered = [None, None]  # per output buffer set: event after its last gather (N > 1)

    def step():
        k = step_no[0] & 1
        sh = sharding.shard_of_step(rank, step_no[0], world)
        step_no[0] += 1
        if gathered[k] is not None:
            gathered[k].synchronize()  # the gather that read this buffer set two steps ago has finished
        solver.solve_batch_device(inits[sh], out_traj[k], out_cost[k], out_i[0], out_i[1], out_i[2], out_i[3])
        if world > 1:  # the one exchange of the path: converged trajectories to rank 0
            sharding.gather_to_root(to_wire(out_traj[k]), sizes)
            sharding.gather_to_root(to_wire(out_cost[k]), sizes)
            gathered[k] = torch.cuda.Event()
            gathered[k].record()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Not a warm-up step and not timed: bring the host core and the GPU out of their idle clocks.  The host
    # keeps the stream two rounds ahead of the device; on a freshly started process its first 100 ms can be slow
    # enough (one run in a dozen, fresh box) for the device to wait on launches during the 70 ms timed region.
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        solver.solve_batch_device(init, out_traj[0], out_cost[0], out_i[0], out_i[1], out_i[2], out_i[3])
    for _ in range(args.warmup):
        step()
    fence()
    # The warm-up runs with both candidates for dominant kernel timed; the timed region keeps the events
    # on the dominant one only (a timed dispatch carries a completion signal: fewer of them, less
    # perturbation of the rounds being measured).
    calib = solver.profile_get()
    if not args.no_profile and not args.profile_all and args.warmup > 0:
        solver.profile_mode((3 if calib["backward_ms"] >= calib["rollout_ms"] else 4) | (args.event_stride << 8))
    solver.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = to_wire(torch.tensor([dt], dtype=torch.float64, device=dev))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    prof = solver.profile_get()

    status, iters, n_bwd, n_fwd = (t.cpu().numpy() for t in out_i)
    total = B * world * args.steps
    value = total / dt

    if rank == 0:
        # ---- roofline of the dominant kernel (this rank's launches, timed region only)
        knots_bwd = float(n_bwd.sum()) * N * args.steps
        knots_fwd = float(n_fwd.sum()) * N * args.steps
        # ms / launches: the launches that carried events (every event_stride-th launch of the dominant kernel);
        # seen: all its launches in the timed region.  Work is per launch: total work / all launches.
        kern = {
            "k_backward": dict(ms=prof["backward_ms"], launches=prof["backward_launches"], seen=prof["backward_seen"],
                               flops=FLOP_BWD_KNOT * knots_bwd, bytes=BYTES_BWD_KNOT * knots_bwd),
            "k_rollout": dict(ms=prof["rollout_ms"], launches=prof["rollout_launches"], seen=prof["rollout_seen"],
                              flops=FLOP_FWD_KNOT * knots_fwd, bytes=BYTES_FWD_KNOT * knots_fwd),
        }
        dom = max(kern, key=lambda k: kern[k]["ms"])
        kd = kern[dom]
        # time all launches would take at the measured average launch duration
        sec = max(kd["ms"], 1e-9) * 1e-3 * max(kd["seen"], 1) / max(kd["launches"], 1)
        tflops = kd["flops"] / sec / 1e12
        gbs = kd["bytes"] / sec / 1e9
        # HBM bytes per launch of that kernel from the latest committed PMC summary (profiles/, produced by
        # profiles/run_rocprof.sh: FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as
        # MI355X_MICROARCH.md prescribes for gfx950); null when no summary is present
        traffic, traffic_src = None, None
        import glob
        summaries = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_rocprof_summary.json")))
        if summaries and B == 1024 and N == 100:
            try:
                js = json.load(open(summaries[-1]))
                traffic = (js["FETCH_SIZE"][dom]["bytes_per_launch_corrected"]
                           + js["WRITE_SIZE"][dom]["bytes_per_launch_corrected"])
                traffic_src = os.path.basename(summaries[-1])
            except Exception:
                traffic = None
        # k_backward is matrix-core work (fp64 MFMA); k_rollout has none: its bound is the bytes it moves
        if dom == "k_backward":
            bound = dict(bound="mfma", achieved=tflops, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s", frac=tflops / FP64_PEAK_TFLOPS)
        else:
            bound = dict(bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS)
        roofline = {
            "kernel": dom, **bound, "traffic": traffic, "traffic_source": traffic_src,
            "avg_launch_us": kd["ms"] * 1e3 / max(kd["launches"], 1), "launches": kd["seen"],
            "timed_launches": kd["launches"],
            "alg_flops_per_launch": kd["flops"] / max(kd["seen"], 1),
            "alg_bytes_per_launch": kd["bytes"] / max(kd["seen"], 1),
            "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS},
            "flops": {"achieved": tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_PEAK_TFLOPS},
            "kernels_ms": {k: round(v["ms"], 3) for k, v in kern.items()}
                          | {"k_linearize": round(prof["linearize_ms"], 3), "other": round(prof["other_ms"], 3)},
            "warmup_avg_launch_us": {k: round(calib[k + "_ms"] * 1e3 / max(calib[k + "_launches"], 1), 2)
                                     for k in ("backward", "rollout")},
        }
        # ---- CPU baseline: the oracle on this host's cores, bounded sample of the same workload
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            from oracle import oracle as orc
            cores = max(1, min(os.cpu_count() or 1, 64))
            sample = min(B, max(64, 24 * cores))
            ref = orc.OracleSolver(orc.model_params(**cfg["model"]), cfg["Q"], cfg["R"], cfg["desired"],
                                   cfg["dt"], orc.options(**cfg["options"]))
            # repeat the sample until about 15 core-seconds of CPU work have been timed
            reps, tc = 0, 0.0
            while reps == 0 or (tc * cores < 15.0 and reps < 64):
                t1 = time.perf_counter()
                r = ref.solve_batch(cfg["init"][:sample], n_threads=cores)
                tc += time.perf_counter() - t1
                reps += 1
            t1 = time.perf_counter()
            ref.solve_batch(cfg["init"][:16], n_threads=1)
            t1c = time.perf_counter() - t1
            got = out_cost[(step_no[0] - 1) & 1].cpu().numpy()[:sample]
            cpu = {"value": sample * reps / tc, "unit": "solves/s", "cores": cores, "kind": "port",
                   "sample": f"first {sample} of the {B} problems of rank 0 x {reps} repeats, {cores} threads, {tc:.2f} s; "
                             f"single thread: {16 / t1c:.1f} solves/s on 16 problems",
                   "parity_max_rel_cost_err": float(np.max(np.abs(got - r["cost"]) / np.abs(r["cost"])))}
        # ---- extra, outside the timed region and never `value`: a stream of such batches with several in flight
        # (one solver handle and one host thread per batch in flight): the tail of one batch -- a few trajectories
        # still iterating on an almost idle chip -- overlaps the head of the next
        serving = None
        if not args.no_serving and world == 1:
            import threading
            serving = {"what": f"solves/s over {args.serving_batches} batches of {B} with k batches in flight "
                               "(independent handles, streams and host threads); k = 1 is `value`'s configuration"}
            for k in (2, 3):
                ws = []
                for _ in range(k):
                    sv = capi.from_config(cfg, device=dev.index, sync_every=args.sync_every)
                    bufs = (torch.empty_like(init), torch.empty(B, dtype=torch.float64, device=dev),
                            [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)])
                    sv.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
                    ws.append((sv, bufs))
                per = max(1, args.serving_batches // k)

                def drive(w):
                    for _ in range(per):
                        w[0].solve_batch_device(init, w[1][0], w[1][1], *w[1][2])

                torch.cuda.synchronize()
                t1 = time.perf_counter()
                th = [threading.Thread(target=drive, args=(w,)) for w in ws]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                torch.cuda.synchronize()
                serving[f"in_flight_{k}"] = per * k * B / (time.perf_counter() - t1)
                for w in ws:
                    w[0].close()
        line = {
            "metric": "iLQR solves/sec (batch, 100-knot SE(3) quadrotor)", "value": value, "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[1]: batch={B}/GPU random SE(3) starts -> hover, "
                                   f"{N} knots, fp64, model A, seed 2", "batch_per_gpu": B, "knots": N,
                       "parallelism": f"batch-shard x{world}" + (" + RCCL gather to rank 0" if world > 1 else ""),
                       "shard_assignment": ("one shard" if world == 1 else
                                            f"{world} shards of {B} distinct problems per step; rank r solves shard (r + step) mod {world}")},
            "iters_mean": float(iters.mean()), "iters_max": int(iters.max()),
            "status_counts": np.bincount(status, minlength=4).tolist(),
            "knot_steps_per_s": float((n_bwd.sum() + n_fwd.sum()) * N * world * args.steps / dt),
            "roofline": roofline, "cpu_baseline": cpu, "serving": serving,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
