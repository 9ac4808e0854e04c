#!/usr/bin/env python3
"""bench.py -- iLQR solves/sec.

Default (what the driver runs): BASELINE.json configs[1] -- B = 1024 random SE(3) starts per GPU, 100 knots, fp64,
model A.  One "step" = one batched solve of the whole per-GPU batch, inputs already resident
in HBM.  N > 1: launched by torch.distributed.run, one rank per GPU; a step solves N x 1024 DISTINCT problems (shard k =
problems k*1024 .. (k+1)*1024 of the configs[1] generator; rank r solves shard (r + step) mod N) -- weak scaling, no
data-path collective, and the converged trajectories are gathered on rank 0 over RCCL inside the timed region.  The
line also carries `shard_rounds` (the rollouts of every shard's slowest problem: the straggler that sets a shard's
time) and, from a short leg after the timed region in which every rank solves shard 0 (--shards same as a
diagnostic), `same_shard.machine_efficiency`: the machine's own scaling with the per-GPU work exactly fixed.

--config 3: BASELINE.json configs[3] as specified -- ONE batch of 65536 problems (seed 4) cut into contiguous shards
over the N ranks (8192 per GPU at N = 8; strong scaling: the same total batch at every N), RCCL gather of the
converged trajectories to rank 0 inside the timed region, its time also reported on its own.

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline      the dominant kernel, from HIP events attached to its dispatches inside the timed region (algorithmic
                flops / bytes from BASELINE.md section 4 times the measured pass counts)
  cpu_baseline  the CPU oracle timed on this host's cores on a bounded sample of the same workload (N = 1 only)
  host_to_host  the metric as SURVEY.md section 8(d) words it: host buffers in -> host buffers out through
                qilqr_solve_batch (pinned buffers, PCIe copies included); measured after the timed region, never `value`
  large_batch   the same solver at B = 8192 per GPU (the shard of configs[3]): solves/s and, per kernel, the average
                launch time and the fractions of the fp64 and HBM peaks -- the saturated-machine view
  serving       several batches in flight (independent handles): measured after the timed region, never `value`
  single_solve  the reference's own call pattern (quadrotor_ilqr_binding.cc:34-41, quadrotor_ilqr.py:306): ONE problem per
                call -- BASELINE.json configs[0] (100 knots, model D, initial = desired) through src.quadrotor_ilqr_binding with
                protobuf messages, populate_debug on (the reference's default) and off, beside the oracle on one host core
  reference_faithful  configs[1] on the kernel that evaluates ilqr.hh:126-133 in the reference's own forms (force_general = 1:
                Eigen's pivoted LDL^T, unsymmetrised V_xx) and one non-symmetric-Q run (cost.hh:30-34 allows it)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  This process holds more streams
# than that (torch's, one per solver handle, two per split batch, one per batch in flight), and two streams that land on one
# queue run one after the other: the B = 8192 leg then takes 29.8 ms instead of 22.0 (its two sub-batches serialised).
# Set before the runtime starts; a value already in the environment wins.  `value`'s own region uses one stream.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector = matrix peak (AMD CDNA4 datasheet)
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8 TB/s
# algorithmic work per knot
FLOP_BWD_KNOT, FLOP_FWD_KNOT = 30000.0, 1250.0
BYTES_BWD_KNOT, BYTES_FWD_KNOT, BYTES_IO_KNOT = 86 * 8.0, 103 * 8.0, 53 * 8.0
# The PMC summary `roofline.traffic` is read from: named explicitly, and only used when its tag is this round's
# (profiles/run_rocprof.sh <tag> writes profiles/<tag>_rocprof_summary.json from the same bench command).
ROUND_TAG = "r06"
TRAFFIC_SUMMARY = os.path.join(ROOT, "profiles", "r06g_rocprof_summary.json")


def kernel_table(prof, n_bwd_knots, n_fwd_knots, solves=None):
    """per-kernel work and time of the timed region (HIP events attached to the dispatches).  With the persistent solve
    (k_solve4: the whole solve is one launch) there is one kernel, and its work is all the backward and forward knots."""
    if prof.get("solve_launches", 0) > 0:
        return {"k_solve4": dict(ms=prof["solve_ms"], launches=prof["solve_launches"], seen=prof["solve_seen"],
                                 flops=FLOP_BWD_KNOT * n_bwd_knots + FLOP_FWD_KNOT * n_fwd_knots,
                                 bytes=BYTES_BWD_KNOT * n_bwd_knots + BYTES_FWD_KNOT * n_fwd_knots)}
    if prof.get("rollout_launches", 0) == 0 and prof.get("backward_launches", 0) > 0:
        # k_backward_rollout: the backward pass and the rollout of a round in one launch (one block of four trajectories per CU:
        # up to 1024 trajectories) -- its work is both passes' knots, its time both serial chains
        # k_round (round 4): the same launch with the linearisation of the block's candidates behind the rollout -- then k_linearize
        # is launched once per solve (the initial trajectory), not once per round; a k_round launch holds up to four rounds
        lin = prof.get("linearize_seen", 0)
        name = "k_round" if (lin <= solves if solves else 4 * lin < prof["backward_seen"]) else "k_backward_rollout"
        return {name: dict(ms=prof["backward_ms"], launches=prof["backward_launches"], seen=prof["backward_seen"],
                                           flops=FLOP_BWD_KNOT * n_bwd_knots + FLOP_FWD_KNOT * n_fwd_knots,
                                           bytes=BYTES_BWD_KNOT * n_bwd_knots + BYTES_FWD_KNOT * n_fwd_knots)}
    return {
        "k_backward": dict(ms=prof["backward_ms"], launches=prof["backward_launches"], seen=prof["backward_seen"],
                           flops=FLOP_BWD_KNOT * n_bwd_knots, bytes=BYTES_BWD_KNOT * n_bwd_knots),
        "k_rollout": dict(ms=prof["rollout_ms"], launches=prof["rollout_launches"], seen=prof["rollout_seen"],
                          flops=FLOP_FWD_KNOT * n_fwd_knots, bytes=BYTES_FWD_KNOT * n_fwd_knots),
    }


def rates(kd):
    """achieved TFLOP/s and GB/s of one kernel: its algorithmic work / (all its launches x the measured average
    launch duration) -- events sample every k-th launch"""
    sec = max(kd["ms"], 1e-9) * 1e-3 * max(kd["seen"], 1) / max(kd["launches"], 1)
    return kd["flops"] / sec / 1e12, kd["bytes"] / sec / 1e9


def read_traffic(dom, B, N):
    """HBM bytes per launch of the dominant kernel from this round's committed PMC summary (FETCH_SIZE and WRITE_SIZE in
    separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); None when the summary is
    missing, is not this round's, or was taken on another workload"""
    if not (B == 1024 and N == 100) or not os.path.exists(TRAFFIC_SUMMARY):
        return None, None
    try:
        js = json.load(open(TRAFFIC_SUMMARY))
        if not str(js.get("tag", "")).startswith(ROUND_TAG):
            return None, None
        t = js["FETCH_SIZE"][dom]["bytes_per_launch_corrected"] + js["WRITE_SIZE"][dom]["bytes_per_launch_corrected"]
        return t, os.path.basename(TRAFFIC_SUMMARY) + " (tag " + js["tag"] + ")"
    except Exception:
        return None, None


def single_solve_leg(device, args):
    """BASELINE.json configs[0] as the reference runs it (quadrotor_ilqr.py:256-306 at horizon_s = 10: 100 knots, model D,
    initial = desired, default options) through src.quadrotor_ilqr_binding.QuadrotorILQR.solve with protobuf messages --
    populate_debug on (the reference's default) and off -- and the same problem on the oracle, one host core."""
    import src.demo as demo
    from src.quadrotor_ilqr_binding import QuadrotorILQR
    from quadrotorilqr_amd import problems as pb
    from oracle import oracle as orc
    c0 = pb.config1(10.0)
    m = c0["model"]
    desired = demo.trajectory_message(c0["desired"])
    out = {"what": "BASELINE.json configs[0]: one problem per call, 100 knots, model D (demo constants), initial = desired; "
                   "QuadrotorILQR.solve through the pybind11 surface with protobuf messages in and out; median of 5 calls"}
    for key, dbg in (("populate_debug_on", True), ("populate_debug_off", False)):
        ilqr = QuadrotorILQR(m["mass_kg"], m["inertia"], m["arm_length_m"], m["torque_to_thrust_ratio_m"], m["g_mpss"],
                             c0["Q"], c0["R"], desired, c0["dt"], demo.options_message(dict(c0["options"], populate_debug=dbg)))
        ilqr.solve(desired)
        ts = []
        for _ in range(5):
            t1 = time.perf_counter()
            traj_msg, debug = ilqr.solve(desired)
            ts.append(time.perf_counter() - t1)
        out[key] = {"ms_per_solve": float(np.median(ts)) * 1e3, "debug_entries": len(debug.iter_debugs)}
        if dbg:
            out["final_cost"] = float(debug.iter_debugs[-1].cost) if len(debug.iter_debugs) else None
    # the same two calls at the C ABI (qilqr_solve through ctypes, arrays in and out): what the solver itself pays for the
    # debug capture, without the protobuf encoding and parsing of 100 x 100 knots that the binding -- like the reference's
    # pybind11_protobuf conversion -- adds on top
    from quadrotorilqr_amd import capi
    abi = {}
    for key, dbg in (("populate_debug_on", True), ("populate_debug_off", False)):
        sv = capi.from_config(dict(c0, options=dict(c0["options"], populate_debug=dbg)), device=device)
        sv.solve(c0["desired"])
        ts = []
        for _ in range(5):
            t1 = time.perf_counter()
            _, info = sv.solve(c0["desired"])
            ts.append(time.perf_counter() - t1)
        abi[key] = {"ms_per_solve": float(np.median(ts)) * 1e3, "debug_entries": int(len(info["debug_costs"]))}
        sv.close()
    abi["debug_on_over_off"] = abi["populate_debug_on"]["ms_per_solve"] / abi["populate_debug_off"]["ms_per_solve"]
    out["c_abi"] = abi
    ref = orc.OracleSolver(orc.model_params(**m), c0["Q"], c0["R"], c0["desired"], c0["dt"], orc.options(**c0["options"]))
    ts = []
    for _ in range(3):
        t1 = time.perf_counter()
        r = ref.solve(c0["desired"], debug=True)
        ts.append(time.perf_counter() - t1)
    out["cpu_oracle_one_core"] = {"ms_per_solve": float(np.median(ts)) * 1e3, "iters": int(r["iters"]), "n_fwd": int(r["n_fwd"]),
                                  "final_cost": float(r["cost"])}
    out["debug_on_over_off"] = out["populate_debug_on"]["ms_per_solve"] / out["populate_debug_off"]["ms_per_solve"]
    out["gpu_over_cpu_time"] = out["populate_debug_on"]["ms_per_solve"] / out["cpu_oracle_one_core"]["ms_per_solve"]
    return out


def reference_faithful_leg(cfg, dev, default_cost, args):
    """configs[1] with force_general = 1 -- Q_uu by Eigen's diagonally pivoted LDL^T, V_x = Q_x - K^T Q_uu k and
    V_xx = Q_xx - K^T Q_uu K not symmetrised (ilqr.hh:126-133 as written) -- against the default kernels and the oracle; and
    the same batch with a NON-symmetric Q (cost.hh:30-34 takes any Q), which only this kernel serves."""
    import torch
    from quadrotorilqr_amd import capi
    from oracle import oracle as orc
    B = cfg["init"].shape[0]
    init = torch.from_numpy(cfg["init"]).to(dev)
    bufs = (torch.empty_like(init), torch.empty(B, dtype=torch.float64, device=dev), [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)])
    out = {}
    r = np.random.default_rng(7)
    Qn = cfg["Q"] + 0.05 * np.triu(r.uniform(-1, 1, (12, 12)), 1)  # not symmetric
    for key, c, kw in (("symmetric_weights_force_general_1", cfg, dict(force_general=1)), ("non_symmetric_Q", dict(cfg, Q=Qn), {})):
        sv = capi.from_config(c, device=dev.index, sync_every=args.sync_every, **kw)
        t_settle = time.perf_counter()  # untimed solves first: the legs before this one leave the GPU nearly idle and its clocks low
        while True:
            sv.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
            if (time.perf_counter() - t_settle) * 1e3 >= args.settle_ms:
                break
        reps = 5
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            sv.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
        torch.cuda.synchronize()
        t = (time.perf_counter() - t1) / reps
        cost = bufs[1].cpu().numpy()
        sample = 64
        ref = orc.OracleSolver(orc.model_params(**c["model"]), c["Q"], c["R"], c["desired"], c["dt"],
                               orc.options(**c["options"])).solve_batch(c["init"][:sample], n_threads=usable_cores())
        st = bufs[2][0].cpu().numpy()
        out[key] = {"value": B / t, "unit": "solves/s", "ms_per_solve": t * 1e3,
                    "max_rel_cost_diff_vs_oracle": float(np.max(np.abs(cost[:sample] - ref["cost"]) / np.abs(ref["cost"]))),
                    "oracle_sample": sample,
                    "same_status_iters_as_oracle": int(np.sum((st[:sample] == ref["status"]) & (bufs[2][1].cpu().numpy()[:sample] == ref["iters"]))),
                    "status_counts": np.bincount(st, minlength=4).tolist()}
        if key.startswith("symmetric"):
            out[key]["max_rel_cost_diff_vs_default_kernels"] = float(np.max(np.abs(cost - default_cost) / np.abs(default_cost)))
        else:
            # Q != Q^T at 100 knots has no answer IN THE REFERENCE ALGORITHM (tests/test_oracle_nonsymmetric.py): C_xx = 2 J^T Q J
            # carries Q's antisymmetric part into V_xx at every knot, ilqr.hh:133 (never symmetrised) amplifies it by ~ 1.3 per knot -- a relative
            # asymmetry of 1e-8 already replaces the first knot's gains --, the first, unchecked full step (ilqr.hh:71-73) takes the cost from
            # 1e1..1e2 to 1e21..1e22, and every later Armijo test compares costs of 1e22 whose differences are the rounding of a rollout through
            # those gains.  The exit class is then a property of the ARITHMETIC (the oracle's build without fused multiply-adds, as the
            # reference's .bazelrc builds it: [0, 28, 1, 35] on this sample; the same source with them: [0, 4, 0, 60]; the device, which fuses:
            # [0, 7, 0, 57]), not of the problem ("noise" was the wrong word).  What is arithmetic-independent -- the
            # blow-up, the final cost to a few per cent, one backward pass -- is stated here and held by tests/test_gpu_parity.py::
            # test_non_symmetric_weights_whole_solves, which also holds WHOLE SOLVES with these weights to the oracle at short horizons.
            ref_o = orc.OracleSolver(orc.model_params(**c["model"]), c["Q"], c["R"], c["desired"], c["dt"], orc.options(**c["options"]))
            g, tm = sv.backwards_pass(c["init"][:8])
            dg, dt_ = 0.0, 0.0
            for b in range(8):
                go, to = ref_o.backwards_pass(c["init"][b])
                dg = max(dg, float(np.max(np.abs(g[b] - go)) / np.max(np.abs(go))))
                dt_ = max(dt_, float(np.max(np.abs(tm[b] - to)) / max(float(np.max(np.abs(to))), 1e-300)))  # (over the larger term: one of the two can be 0)
            out[key]["one_backward_pass_vs_oracle"] = {"problems": 8, "max_gain_diff_over_largest_gain": dg, "max_rel_diff_of_cost_reduction_terms": dt_}
            out[key]["oracle_status_counts_on_sample"] = np.bincount(ref["status"], minlength=4).tolist()
            out[key]["status_counts_on_sample"] = np.bincount(st[:sample], minlength=4).tolist()
            fast_L = orc.fast_library(native=False)
            ref_f = orc.OracleSolver(orc.model_params(**c["model"]), c["Q"], c["R"], c["desired"], c["dt"], orc.options(**c["options"]),
                                     library=fast_L).solve_batch(c["init"][:sample], n_threads=usable_cores())
            out[key]["oracle_fma_build_status_counts_on_sample"] = np.bincount(ref_f["status"], minlength=4).tolist()
            out[key]["max_rel_cost_diff_vs_oracle_fma_build"] = float(np.max(np.abs(cost[:sample] - ref_f["cost"]) / np.abs(ref_f["cost"])))
            out[key]["oracle_builds_max_rel_cost_diff"] = float(np.max(np.abs(ref["cost"] - ref_f["cost"]) / np.abs(ref["cost"])))
            out[key]["median_final_cost"] = float(np.median(cost[:sample]))
            out[key]["note"] = ("Q != Q^T at 100 knots: ilqr.hh:133 amplifies Q's antisymmetric part by ~1.3 per knot, the first unchecked step takes the cost to "
                                "~1e21, and the exit class (1: a step too small to move the cost; 3: search exhausted) is decided by the arithmetic's rounding -- the "
                                "oracle's builds with and without fused multiply-adds differ from each other as the device differs from either "
                                "(tests/test_oracle_nonsymmetric.py); parity is stated for one backward pass, for the blow-up and the final cost, and for whole solves "
                                "at short horizons (tests/test_gpu_parity.py::test_non_symmetric_weights_whole_solves)")
        sv.close()
    out["what"] = (f"B = {B}, N = {cfg['init'].shape[1]}, device-resident, 5 repeats behind {args.settle_ms:.0f} ms of untimed solves: k_backward<false> (one wavefront "
                   "per trajectory, dense records, Eigen's pivoted LDL^T, the reference's unsymmetrised V_xx) with the default rollout and linearisation kernels")
    return out


def sharded_c_abi_child(args):
    """The N > 1 line's step through the PRODUCT's own multi-GPU entry point, qilqr_solve_batch_sharded_device (north_star: "host side
    is C++ ... RCCL-over-xGMI gather"): ONE process, one solver, stream and host thread per device, every shard's rows gathered into
    device arrays on shard 0's device by the library's transport (RCCL groups per shard when the shards sit on different devices).
    Runs in a child process of rank 0 after the timed region, the other ranks idle at a barrier (never `value`); a child, so
    that a communicator that fails to come up on hardware nobody has run it on ends a leg, not the line.  Host (pinned) inputs ->
    device-resident gathered outputs.  Prints one JSON object; writes the gathered costs to --leg-out for the parent's parity check."""
    import torch
    from quadrotorilqr_amd import capi, problems as pb
    n_dev, N = args.gpus, args.knots
    one_device = os.environ.get("QILQR_BENCH_ONE_DEVICE_TEST") == "1"
    devices = [0] * n_dev if one_device else list(range(n_dev))
    if args.config == 3:
        B_total, seed = args.batch or 65536, 4
    else:
        B_total, seed = (args.batch or 1024) * n_dev, 2
    cfg = pb.config2(B=B_total, N=N, seed=seed)  # counter-based generator: the N shards of the rank-per-GPU run, in global order
    many = capi.sharded_from_config(cfg, devices=devices, sync_every=args.sync_every)
    transport = many.set_transport("auto")
    hin = capi.host_array(cfg["init"].shape)
    hin[...] = cfg["init"]
    root = torch.device("cuda", devices[0])
    outs = (torch.empty((B_total, N, 18), dtype=torch.float64, device=root), torch.empty(B_total, dtype=torch.float64, device=root),
            *[torch.empty(B_total, dtype=torch.int32, device=root) for _ in range(4)])
    steps = max(2, min(args.steps, 10))
    for _ in range(2):
        many.solve_batch_gathered(hin, *outs, root=0)
    torch.cuda.synchronize(root)
    gms = []
    t1 = time.perf_counter()
    for _ in range(steps):
        gms.append(many.solve_batch_gathered(hin, *outs, root=0))
    torch.cuda.synchronize(root)
    dt = (time.perf_counter() - t1) / steps
    cost = outs[1].cpu().numpy()
    if args.leg_out:
        np.save(args.leg_out, cost)
    st = outs[2].cpu().numpy()
    print(json.dumps({"value": B_total / dt, "unit": "solves/s", "ms_per_step": dt * 1e3, "steps": steps, "devices": devices,
                      "transport": transport, "exposed_gather_ms": float(np.median(gms)), "batch_total": B_total,
                      "status_counts": np.bincount(st, minlength=4).tolist(),
                      "what": "qilqr_solve_batch_sharded_device: one process, one solver / stream / host thread per device, pinned host "
                              "inputs, results gathered on shard 0's device by the library's own transport"}))
    many.close()


def sharded_c_abi_leg(args, g_cost_host):
    """rank 0, after the timed region: run sharded_c_abi_child in a child process (bounded), compare its gathered costs with the
    rank-per-GPU path's last gathered step"""
    import subprocess
    import tempfile
    tmp = tempfile.NamedTemporaryFile(suffix=".npy", delete=False)
    tmp.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                           "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    cmd = [sys.executable, os.path.abspath(__file__), "--leg", "sharded-c-abi", "--leg-out", tmp.name, "--gpus", str(args.gpus),
           "--steps", str(args.steps), "--config", str(args.config), "--batch", str(args.batch), "--knots", str(args.knots),
           "--sync-every", str(args.sync_every)]
    try:
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=150)
        except subprocess.TimeoutExpired:
            return {"error": "the child process did not finish within 150 s (killed)"}
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or len(lines) != 1:
            return {"error": f"child exit code {r.returncode}", "stderr_tail": r.stderr[-600:]}
        out = json.loads(lines[0])
        if g_cost_host is not None:
            got = np.load(tmp.name)
            same = got.shape == g_cost_host.shape and bool(np.array_equal(got, g_cost_host))
            out["same_costs_as_rank_per_gpu_gather"] = same
            if not same and got.shape == g_cost_host.shape:
                out["max_rel_cost_diff_vs_rank_per_gpu_gather"] = float(np.max(np.abs(got - g_cost_host) / np.abs(g_cost_host)))
        return out
    finally:
        try:
            os.unlink(tmp.name)
        except OSError:
            pass


def usable_cores():
    """hardware threads this process may actually use: the affinity mask cut by the cgroup's CPU quota (a GPU box hands a one-GPU
    job 16 of the host's cores; os.cpu_count() reports all of them)"""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = float(txt[0])
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return max(1, n)


def launch_ranks_if_needed(args):
    """`python3 bench.py --gpus N` with no launcher IS an N-rank run: when N > 1 and no rank environment is present, start
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <same arguments>` as a CHILD process (nothing in
    this process has touched the GPU, and it never execs), let its rank 0 print the line on the inherited stdout, and leave with
    its return code.  A rank environment that contradicts --gpus is an error, not a silently different run."""
    in_launcher = "RANK" in os.environ or "WORLD_SIZE" in os.environ
    if in_launcher:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if args.gpus is None:  # `torchrun --nproc-per-node N bench.py` without --gpus: the launcher says how many
            args.gpus = world
        if world != args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher's WORLD_SIZE is {world}\n")
            sys.exit(2)
        return
    if args.gpus is None:
        args.gpus = 1
    if args.gpus <= 1:
        return
    import socket
    import subprocess
    # preflight: are there N devices?  Asked in a throw-away CHILD process -- this process must not touch the GPU (it
    # starts the ranks and never execs) -- before N ranks are started that would each fail on their own with a stack trace.
    if os.environ.get("QILQR_BENCH_ONE_DEVICE_TEST") != "1":
        try:
            r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
            have = int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else -1
        except Exception:
            have = -1
        if 0 <= have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but this machine shows {have} GPU(s) (torch.cuda.device_count() in a child process); "
                             f"nothing was started\n")
            sys.exit(2)
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="GPUs = ranks of one node (default: the launcher's WORLD_SIZE when there is one, else 1)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--shards", choices=("same", "distinct"), default="distinct",
                    help="config 1 with N > 1: 'distinct' (default) = a step is N x 1024 distinct problems, rank r solves shard (r + step) mod N; "
                         "'same' = diagnostic: every rank solves shard 0 (the problems of the N = 1 line: per-GPU work exactly fixed)")
    ap.add_argument("--config", type=int, default=1, choices=(1, 3),
                    help="BASELINE.json configs[k]: 1 = B 1024 per GPU (weak scaling, default); 3 = one batch of 65536 sharded over the GPUs (strong scaling)")
    ap.add_argument("--batch", type=int, default=0, help="config 1: problems per GPU (default 1024); config 3: problems in total (default 65536)")
    ap.add_argument("--knots", type=int, default=100)
    ap.add_argument("--sync-every", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-all", action="store_true", help="HIP events around every kernel, not only the two candidates for dominant kernel")
    ap.add_argument("--rollout", type=int, default=-1, help="rollout kernel (qilqr_device_config.single_wave_rollout): 0 by the batch (k_rollout16 up to 4096 trajectories, k_rollout3 beyond), "
                                                             "1 k_rollout, 2 k_rollout3, 3 k_rollout16 (default: library default)")
    ap.add_argument("--backward", type=int, default=0, help="diagnostic: backward kernel (qilqr_device_config.force_general: 0 automatic, 1 general, 2 one wavefront per trajectory, 3 k_backward2 (diagnostics build), 4 k_backward4 six wavefronts, 5 fused)")
    ap.add_argument("--persistent", type=int, default=0, help="qilqr_device_config.persistent: 0 / 2 rounds of three launches (the product), 1 the solve as one launch (k_solve4: loads the diagnostics build of the library)")
    ap.add_argument("--streams", type=int, default=0, help="sub-batches on their own streams (qilqr_device_config.streams; 0 automatic)")
    ap.add_argument("--event-stride", type=int, default=0, help="time every k-th launch of the dominant kernel in the timed region (a timed dispatch costs the stream about 6 us: "
                    "with every 4th launch of ~115 us timed `value` reads 1.4 %% low, with every 16th 0.2 %% -- profiles/r04_event_stride.txt); 0 = by the "
                    "launch time seen in the warm-up: every 5th when a launch holds several rounds (> 300 us), every 16th otherwise")
    ap.add_argument("--settle-ms", type=float, default=300.0, help="untimed solves before the warm-up steps (clocks out of idle)")
    ap.add_argument("--no-serving", action="store_true", help="skip the extra several-batches-in-flight measurement (never part of value)")
    ap.add_argument("--serving-batches", type=int, default=18)
    ap.add_argument("--no-host-to-host", action="store_true", help="skip the host-buffers-in / host-buffers-out measurement (never part of value)")
    ap.add_argument("--no-large-batch", action="store_true", help="skip the B = 8192 measurement (never part of value)")
    ap.add_argument("--rehearse-nccl", action="store_true",
                    help="N = 1 only: initialise torch.distributed with the REAL nccl (RCCL) backend at world size 1 and send every step's results "
                         "through gather_to_root's send / receive pair addressed to rank 0 itself -- the N > 1 line's communication calls, "
                         "communicator creation and two-HIP-runtimes process on the one GPU of a test box (never the driver's default run)")
    ap.add_argument("--no-single-solve", action="store_true", help="skip the one-problem-per-call measurement through the binding (never part of value)")
    ap.add_argument("--no-reference-faithful", action="store_true", help="skip the force_general = 1 / non-symmetric-Q measurement (never part of value)")
    ap.add_argument("--no-profile", action="store_true", help="diagnostic: no HIP events around the kernels (roofline = null)")
    ap.add_argument("--no-sharded-c-abi", action="store_true", help="N > 1: skip the extra measurement of the same step through qilqr_solve_batch_sharded_device "
                                                                   "(one process, N devices, the library's own RCCL gather; never part of value)")
    ap.add_argument("--leg", default="", help=argparse.SUPPRESS)  # internal: a leg bench.py runs in a child process of rank 0 ("sharded-c-abi")
    ap.add_argument("--leg-out", default="", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.leg == "sharded-c-abi":
        return sharded_c_abi_child(args)
    launch_ranks_if_needed(args)

    import torch
    import torch.distributed as dist

    if args.persistent == 1 or args.backward == 3:  # k_solve4 / k_backward2 live in the diagnostics build (never the driver's default run)
        os.environ.setdefault("QILQR_LIB", os.path.join(ROOT, "quadrotorilqr_amd", "lib", "libquadrotor_ilqr_diag.so"))
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
        if args.rehearse_nccl:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda", local_rank if (world > 1 and not one_device_test) else 0)
    to_wire = (lambda t: t.cpu()) if one_device_test else (lambda t: t)  # gloo gathers host tensors

    N = args.knots
    strong = args.config == 3
    if strong:
        # configs[3]: ONE batch, contiguous shards, fixed assignment (rank r holds [lo_r, hi_r) of the global batch)
        B_total = args.batch or 65536
        sizes = sharding.shard_sizes(B_total, world)
        lo, hi = sharding.shard_range(B_total, rank, world)
        B, seed = hi - lo, 4
        cfg = pb.config2(B=B, N=N, seed=seed, b0=lo)  # counter-based generator: keyed by the global problem index
        shard_at = lambda step: list(range(world))
    else:
        # configs[1]: 1024 problems per GPU, weak scaling.  How long a batch takes is set by its slowest problem, and the
        # 1024-problem shards of one seeded sequence differ in that (the first eight: 30 to 45 rollouts, 4.96 to 6.81 ms,
        # shard 0 -- the N = 1 line -- 6 % faster than their mean: profiles/microbench/shard_times.py).  That spread IS
        # the workload ("random SE(3) starts per GPU"), so it is measured, and shown in `shard_rounds`:
        #   --shards distinct (default): the global batch of a step is N shards of 1024 distinct problems (shard k =
        #     problems k*B .. (k+1)*B of the generator) and rank r solves shard (r + step) mod N: a fixed assignment would
        #     make every step wait for the same unlucky rank, rotating it evens the ranks' totals over the steps without
        #     any exchange (sharding.shard_of_step);
        #   --shards same (diagnostic): every rank solves shard 0 -- the per-GPU work is exactly that of the N = 1 line.
        #     The default run measures this too, in a short leg after the timed region (`same_shard`).
        # (One batch of distinct problems cut over the GPUs is --config 3.)
        B, seed = args.batch or 1024, 2
        B_total = B * world
        sizes = [B] * world
        same = args.shards == "same"
        cfg = pb.config2(B=B, N=N, seed=seed, b0=0 if same else rank * B)
        if same:
            shard_at = lambda step: list(range(world))
        else:
            shard_at = lambda step: [sharding.shard_of_step(r, step, world) for r in range(world)]
    solver = capi.from_config(cfg, device=dev.index, profile=(0 if args.no_profile else (2 if args.profile_all else 1)), sync_every=args.sync_every,
                              force_general=args.backward, streams=args.streams, persistent=args.persistent,
                              **({} if args.rollout < 0 else dict(single_wave_rollout=args.rollout)))

    init = torch.from_numpy(cfg["init"]).to(dev)
    inits = {rank: init}  # by shard index
    if not strong and same:
        inits = {sh: init for sh in range(world)}  # every "shard" is shard 0
    if not strong and not same:
        for sh in range(world):
            if sh not in inits:
                inits[sh] = torch.from_numpy(pb.config2(B=B, N=N, seed=seed, b0=sh * B)["init"]).to(dev)
    # two sets of output buffers, used alternately: with N > 1 the gather of step s (RCCL, torch's stream) runs
    # while the solver's own stream is already solving step s + 1
    out_traj = [torch.empty_like(init) for _ in range(2)]
    out_cost = [torch.empty(B, dtype=torch.float64, device=dev) for _ in range(2)]
    out_i = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)]  # status, iters, n_bwd, n_fwd
    # the gathered batch on rank 0, in global problem order, allocated once
    rehearse = world == 1 and args.rehearse_nccl
    if (world > 1 or rehearse) and rank == 0:
        wire_dev = torch.device("cpu") if one_device_test else dev
        g_traj = torch.empty((B_total, N, 18), dtype=torch.float64, device=wire_dev)
        g_cost = torch.empty(B_total, dtype=torch.float64, device=wire_dev)
    else:
        g_traj = g_cost = None
    step_no = [0]
    gathered = [None, None]  # per output buffer set: event after its last gather (N > 1)
    pass_knots = torch.zeros(2, dtype=torch.float64, device=dev)  # sum over the timed steps of n_bwd, n_fwd (this rank)
    # per shard: the rollouts of its slowest problem (= the rounds its batch solve takes), as seen in the timed steps
    shard_rounds = torch.zeros(world, dtype=torch.int32, device=dev)
    count_passes = [False]

    def gather(k, step):
        sor = shard_at(step)
        sharding.gather_to_root(to_wire(out_traj[k]), sizes, out=g_traj, shard_of_rank=sor, rehearse_self=rehearse)
        sharding.gather_to_root(to_wire(out_cost[k]), sizes, out=g_cost, shard_of_rank=sor, rehearse_self=rehearse)

    def step():
        k = step_no[0] & 1
        sh = shard_at(step_no[0])[rank]
        if gathered[k] is not None:
            gathered[k].synchronize()  # the gather that read this buffer set two steps ago has finished
        # (static inputs; the gather still in flight on torch's stream reads the OTHER buffer set: no ordering needed)
        solver.solve_batch_device(inits[sh], out_traj[k], out_cost[k], out_i[0], out_i[1], out_i[2], out_i[3],
                                  wait_current_stream=False)
        if count_passes[0] and world > 1:  # rotating shards: the pass counts differ from step to step
            pass_knots.add_(torch.stack([out_i[2].sum(), out_i[3].sum()]).to(torch.float64))
            shard_rounds[sh] = torch.maximum(shard_rounds[sh], out_i[3].max())
        if world > 1 or rehearse:  # the one exchange of the path: converged trajectories to rank 0
            gather(k, step_no[0])
            gathered[k] = torch.cuda.Event()
            gathered[k].record()
        step_no[0] += 1

    def fence():
        if world > 1 or rehearse:
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
        dom_ms, dom_n = max((calib["backward_ms"], calib["backward_launches"]), (calib["rollout_ms"], calib["rollout_launches"]))
        # (strides that share no factor with the launches of a solve -- 37 of one round, 12 or 13 of four -- so that the sample covers every
        # position of a solve: with every 4th of 12 timed, always the same three launches, the average read 440 us for rocprofv3's 382)
        stride = args.event_stride or (5 if dom_ms * 1e3 / max(dom_n, 1) > 300.0 else 16)
        solver.profile_mode((3 if calib["backward_ms"] >= calib["rollout_ms"] else 4) | (stride << 8))  # (k_solve4 is timed in every mode)
    solver.profile_reset()
    count_passes[0] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    count_passes[0] = False
    if world > 1:
        tt = to_wire(torch.tensor([dt], dtype=torch.float64, device=dev))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    prof = solver.profile_get()

    # the gather on its own (after the timed region): every rank's shard of the last step to rank 0, waited for
    gather_ms = None
    if world > 1:
        fence()
        t1 = time.perf_counter()
        for _ in range(3):
            gather((step_no[0] - 1) & 1, step_no[0] - 1)
        fence()
        tg = to_wire(torch.tensor([(time.perf_counter() - t1) / 3], dtype=torch.float64, device=dev))
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gather_ms = float(tg.item()) * 1e3
    # (rank 0 now holds the last timed step's costs of the WHOLE batch in global problem order: the comparand of the sharded_c_abi leg)
    ref_cost = None
    if world > 1 and rank == 0 and (strong or args.shards != "same"):
        ref_cost = g_cost.cpu().numpy().copy()

    # ---- diagnostic leg (N > 1, configs[1], after the timed region, never `value`): the machine's own scaling.  Every rank
    # solves shard 0 -- exactly the N = 1 line's work -- with the gather, all ranks together; then rank 0 solves it alone.
    same_shard = None
    if world > 1 and not strong:
        ks = max(2, min(args.steps, 10))
        sor0 = list(range(world))

        def same_step(k):
            solver.solve_batch_device(inits[0], out_traj[k], out_cost[k], out_i[0], out_i[1], out_i[2], out_i[3],
                                      wait_current_stream=False)
            sharding.gather_to_root(to_wire(out_traj[k]), sizes, out=g_traj, shard_of_rank=sor0)
            sharding.gather_to_root(to_wire(out_cost[k]), sizes, out=g_cost, shard_of_rank=sor0)

        same_step(0)
        fence()
        t1 = time.perf_counter()
        for i in range(ks):
            same_step(i & 1)
        fence()
        ts = to_wire(torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev))
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        alone = None
        if rank == 0:  # (the other ranks wait at the barrier of the fence below, their GPUs idle)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(ks):
                solver.solve_batch_device(inits[0], out_traj[0], out_cost[0], out_i[0], out_i[1], out_i[2], out_i[3],
                                          wait_current_stream=False)
            torch.cuda.synchronize()
            alone = time.perf_counter() - t1
        fence()
        if rank == 0:
            v_same, v_alone = B * world * ks / float(ts.item()), B * ks / alone
            same_shard = {"what": f"every rank solves shard 0 (the N = 1 line's problems) + gather to rank 0, {ks} steps; then rank 0 alone, {ks} steps",
                          "value": v_same, "ms_per_step": float(ts.item()) / ks * 1e3, "one_rank_alone_value": v_alone,
                          "machine_efficiency": v_same / (world * v_alone)}
        # (the last solve of this leg overwrote out_i with shard 0's counts: the line's iters/status are shard 0's)

    sharded_abi = None  # (the product's own multi-GPU path runs LAST, behind the process group's teardown: below)

    status, iters, n_bwd, n_fwd = (t.cpu().numpy() for t in out_i)
    if world > 1:
        sr = to_wire(shard_rounds.clone())
        dist.all_reduce(sr, op=dist.ReduceOp.MAX)
        shard_rounds_list = [int(v) for v in sr.cpu().numpy()]
    else:
        shard_rounds_list = [int(n_fwd.max())]
    total = B_total * args.steps
    value = total / dt
    if world > 1:
        knots_bwd, knots_fwd = (float(v) * N for v in pass_knots.cpu().numpy())
        all_knots = to_wire(pass_knots.clone())
        dist.all_reduce(all_knots, op=dist.ReduceOp.SUM)
        knot_steps = float(all_knots.sum().item()) * N
    else:  # one shard, the same every step
        knots_bwd, knots_fwd = float(n_bwd.sum()) * N * args.steps, float(n_fwd.sum()) * N * args.steps
        knot_steps = knots_bwd + knots_fwd

    if rank == 0:
        # ---- roofline of the dominant kernel (this rank's launches, timed region only)
        roofline = None
        if not args.no_profile:
            kern = kernel_table(prof, knots_bwd, knots_fwd, solves=args.steps)
            dom = max(kern, key=lambda k: kern[k]["ms"])
            kd = kern[dom]
            tflops, gbs = rates(kd)
            traffic, traffic_src = read_traffic(dom, B, N)
            # k_backward (and the persistent solve, which contains it) is matrix-core work (fp64 MFMA); k_rollout has none:
            # its bound is the bytes it moves
            if dom in ("k_backward", "k_backward_rollout", "k_round", "k_solve4"):
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
                **({"note": "k_backward_rollout = the backward pass and the rollout of a round in one launch (up to 1024 trajectories); its "
                            "work is both passes' algorithmic flops, its time both serial chains (the rollout's has no matrix work).  Launched apart "
                            "(qilqr_device_config.round_launch = 1) the backward kernel alone ran at 20.6 % of the fp64 peak in round 3 (68.9 us per launch, "
                            "profiles/r03f_rocprof_summary.txt; 67.2 us with one live wavefront per block since round 4's pipelined knot, "
                            "profiles/r04_knot_anatomy.txt) and the rollout at 51.8 us.  The numerator is the reference's dense-as-written 30 kflop "
                            "per backward knot; the kernel issues 7 x 2048 flop of MFMA + ~68 fp64 vector instructions per knot"} if dom == "k_backward_rollout" else {}),
                **({"note": "k_round = a whole round in one launch (up to 1024 trajectories, a block of four per CU): the block's backward pass, the "
                            "rollout of its four trajectories with the linearisation of the candidates behind it (round 6), serial chains one after the other "
                            "(about 152 + 134 + 11 thousand shader cycles with one running trajectory per block), and up to FOUR such rounds per launch (avg_launch_us is a "
                            "launch of four).  Its work is the backward and forward knots' algorithmic "
                            "flops -- the reference's backward knot, 30 kflop dense-as-written, calls the dynamics and cost differentials itself (ilqr.hh:110-116), which "
                            "until round 4 ran in a launch of its own outside this denominator: the fraction fell from 0.129 (k_backward_rollout, "
                            "114 us per launch) for that reason while `value` rose 2 %.  The backward kernel alone ran at 20.6 % of the fp64 peak in "
                            "round 3 (68.9 us per launch; 67.2 us with one live wavefront per block since the pipelined knot, "
                            "profiles/r04_knot_anatomy.txt); the kernel issues 7 x 2048 flop of MFMA + ~68 fp64 vector instructions per knot"}
                   if dom == "k_round" else {}),
                "kernels_ms": {k: round(v["ms"], 3) for k, v in kern.items()}
                              | {"k_linearize": round(prof["linearize_ms"], 3), "other": round(prof["other_ms"], 3)},
                "warmup_avg_launch_us": {k: round(calib[k + "_ms"] * 1e3 / max(calib[k + "_launches"], 1), 2)
                                         for k in ("backward", "rollout", "solve")},
            }
        # ---- the extra legs (never `value`).  HIP hands hardware queues to streams in the order the streams are created and does
        # not give a destroyed stream's place back, and GPU_MAX_HW_QUEUES = 8: host_to_host is measured FIRST and on `value`'s own
        # handle (one more stream: its copy-back's), then the B = 8192 solver creates its five (streams three to eight of the
        # process: behind further handles they end up sharing queues -- 385 000 instead of 444 000-460 000 solves/s in round 3; and a
        # fresh host_to_host handle BEHIND them read + 0.60 ms over the device-resident call instead of + 0.43, round 6).
        # ---- the metric as SURVEY.md section 8(d) defines it: host buffers in -> host buffers out (never `value`).  Pinned and
        # pageable buffers alternate call by call; median and 90th percentile of >= 30 calls each.
        h2h = None
        if not args.no_host_to_host and world == 1:
            hin = capi.host_array(cfg["init"].shape)
            hin[...] = cfg["init"]
            hout = dict(traj=capi.host_array(cfg["init"].shape), cost=capi.host_array((B,)),
                        **{k: capi.host_array((B,), np.int32) for k in ("status", "iters", "n_bwd", "n_fwd")})
            # On `value`'s own handle, its event timing switched off (the profile of the timed region was read above): a fresh handle's two
            # streams (its own and the copy-back's) would be the ninth and tenth of the process behind the B = 8192 solver's five, and HIP
            # multiplexes streams onto GPU_MAX_HW_QUEUES = 8 hardware queues -- round 6 measured + 0.60 ms over the device-resident call with
            # the fresh handle behind the large solver and + 0.43 without the large solver in the process: a collision of the bench's own
            # making, not the library's (profiles/microbench/h2h_ab.py: + 0.33-0.36 in a process of its own).
            solver.profile_mode(0)
            plain = solver
            pg = cfg["init"].copy()  # the same through pageable buffers (HIP stages the copies itself)
            t_settle = time.perf_counter()
            while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:  # clocks and both paths warm
                plain.solve_batch(hin, out=hout)
                plain.solve_batch(pg)
            reps = max(30, args.steps)
            tpin, tpag, tdev = [], [], []
            for _ in range(reps):
                t1 = time.perf_counter()
                plain.solve_batch(hin, out=hout)
                tpin.append(time.perf_counter() - t1)
                t1 = time.perf_counter()
                plain.solve_batch(pg)
                tpag.append(time.perf_counter() - t1)
                t1 = time.perf_counter()  # and the device-resident call of `value`, same handle state, same moment
                solver.solve_batch_device(init, out_traj[0], out_cost[0], out_i[0], out_i[1], out_i[2], out_i[3])
                tdev.append(time.perf_counter() - t1)
            tpin, tpag, tdev = (np.array(t) * 1e3 for t in (tpin, tpag, tdev))
            th = float(np.median(tpin))
            h2h = {"value": B / th * 1e3, "unit": "solves/s", "ms_per_solve": th, "ms_p90": float(np.percentile(tpin, 90)),
                   "ms_min": float(tpin.min()),
                   "what": f"qilqr_solve_batch, B = {B}: quaternion checks + H2D + solve + D2H, pinned host buffers; median of {reps} calls, "
                           "first of the extra legs, alternating with the pageable-buffer call and the device-resident call",
                   "bytes_in": int(cfg["init"].nbytes), "bytes_out": int(cfg["init"].nbytes + B * 24),
                   "pageable_buffers": {"value": B / float(np.median(tpag)) * 1e3, "ms_per_solve": float(np.median(tpag)),
                                        "ms_p90": float(np.percentile(tpag, 90))},
                   "device_resident_same_moment": {"ms_per_solve": float(np.median(tdev)), "ms_p90": float(np.percentile(tdev, 90))},
                   "over_device_resident_ms": th - float(np.median(tdev)),
                   "parity_with_device_path": bool(np.array_equal(hout["cost"], out_cost[0].cpu().numpy()))}
        large = None
        ls = None
        if not args.no_large_batch and world == 1 and not strong:
            LB = 8192
            lcfg = pb.config2(B=LB, N=N, seed=4)
            ls = capi.from_config(lcfg, device=dev.index, profile=2, sync_every=args.sync_every)
            linit = torch.from_numpy(lcfg["init"]).to(dev)
            lbuf = (torch.empty_like(linit), torch.empty(LB, dtype=torch.float64, device=dev),
                    [torch.empty(LB, dtype=torch.int32, device=dev) for _ in range(4)])
            ls.solve_batch_device(linit, lbuf[0], lbuf[1], *lbuf[2])
            torch.cuda.synchronize()
        # ---- the saturated machine: the shard one GPU solves in configs[3] (never `value`)
        if ls is not None:
            t_settle = time.perf_counter()  # untimed solves first: the legs before this one leave the GPU idle for seconds and its clocks low
            while True:
                ls.solve_batch_device(linit, lbuf[0], lbuf[1], *lbuf[2])
                if (time.perf_counter() - t_settle) * 1e3 >= args.settle_ms:
                    break
            ls.profile_reset()
            reps = 3
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                ls.solve_batch_device(linit, lbuf[0], lbuf[1], *lbuf[2])
            tl = (time.perf_counter() - t1) / reps
            lp = ls.profile_get()
            lb_, lf_ = float(lbuf[2][2].sum().item()) * N * reps, float(lbuf[2][3].sum().item()) * N * reps
            lk = kernel_table(lp, lb_, lf_)
            per = {}
            for k, kd in lk.items():
                if kd["launches"] == 0:
                    continue
                tf, gb = rates(kd)
                per[k] = {"avg_launch_us": kd["ms"] * 1e3 / max(kd["launches"], 1), "launches": kd["launches"],
                          "fp64_frac": tf / FP64_PEAK_TFLOPS, "hbm_frac": gb / HBM_PEAK_GBS}
            if lp["linearize_launches"]:
                per["k_linearize"] = {"avg_launch_us": lp["linearize_ms"] * 1e3 / max(lp["linearize_launches"], 1),
                                      "launches": lp["linearize_launches"]}
            ls.profile_mode(0)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                ls.solve_batch_device(linit, lbuf[0], lbuf[1], *lbuf[2])
            tl0 = (time.perf_counter() - t1) / reps
            large = {"what": f"B = {LB}, N = {N}, fp64, seed 4 (the shard one GPU solves in configs[3]); device-resident, {reps} repeats",
                     "value": LB / tl0, "unit": "solves/s", "ms_per_solve": tl0 * 1e3,
                     "ms_per_solve_with_every_kernel_timed": tl * 1e3, "kernels": per,
                     "status_counts": np.bincount(lbuf[2][0].cpu().numpy(), minlength=4).tolist()}
            ls.close()
            del linit, lbuf
        # ---- CPU baseline: the oracle on this host's cores, bounded sample of the same workload.  Timed on the TIMING build of the
        # oracle's source (fused multiply-adds allowed, compiled for this host when gcc is here: oracle/Makefile `fast-native`), every
        # hardware thread this process may use, problems drawn from an atomic counter; parity is checked on the PARITY build.
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            from oracle import oracle as orc
            cores = usable_cores()
            sample = min(B, max(64, 24 * cores))
            mk = lambda L: orc.OracleSolver(orc.model_params(**cfg["model"]), cfg["Q"], cfg["R"], cfg["desired"],
                                            cfg["dt"], orc.options(**cfg["options"]), library=L)
            fast = mk(orc.fast_library())
            fast.solve_batch(cfg["init"][:cores], n_threads=cores)  # threads, pages and clocks warm
            # repeat the sample until about 15 core-seconds of CPU work have been timed
            reps, tc = 0, 0.0
            while reps == 0 or (tc * cores < 15.0 and reps < 64):
                t1 = time.perf_counter()
                fast.solve_batch(cfg["init"][:sample], n_threads=cores)
                tc += time.perf_counter() - t1
                reps += 1
            t1 = time.perf_counter()
            fast.solve_batch(cfg["init"][:32], n_threads=1)
            t1c = time.perf_counter() - t1
            r = mk(None).solve_batch(cfg["init"][:sample], n_threads=cores)  # the parity build: the checker
            got = out_cost[(step_no[0] - 1) & 1].cpu().numpy()[:sample]
            v = sample * reps / tc
            cpu = {"value": v, "unit": "solves/s", "cores": cores, "kind": "port",
                   "sample": f"first {sample} of the {B} problems of rank 0 x {reps} repeats, {cores} threads (affinity and cgroup quota; the host "
                             f"reports {os.cpu_count()}), problems drawn from an atomic counter, {tc:.2f} s; timing build of the oracle "
                             f"[{fast.flavour()}]; single thread: {32 / t1c:.1f} solves/s on 32 problems",
                   "single_thread_value": 32 / t1c, "thread_scaling": v / (cores * 32 / t1c),
                   "parity_build": mk(None).flavour(),
                   "parity_max_rel_cost_err": float(np.max(np.abs(got - r["cost"]) / np.abs(r["cost"])))}
        # ---- extra, outside the timed region and never `value`: a stream of such batches with several in flight
        # (one solver handle and one host thread per batch in flight): the tail of one batch -- a few trajectories
        # still iterating on an almost idle chip -- overlaps the head of the next
        serving = None
        if not args.no_serving and world == 1 and not strong:
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
                        w[0].solve_batch_device(init, w[1][0], w[1][1], *w[1][2], wait_current_stream=False)

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
        # ---- the reference's own call pattern: one problem per call through the pybind11 surface (never `value`)
        single = None
        if not args.no_single_solve and world == 1 and not strong:
            single = single_solve_leg(dev.index, args)
        # ---- the kernel that evaluates ilqr.hh:126-133 in the reference's own forms (never `value`)
        faithful = None
        if not args.no_reference_faithful and world == 1 and not strong:
            faithful = reference_faithful_leg(cfg, dev, out_cost[(step_no[0] - 1) & 1].cpu().numpy(), args)
        if strong:
            workload = (f"BASELINE.json configs[3]: ONE batch of {B_total} random SE(3) starts -> hover, {N} knots, fp64, model A, "
                        f"seed 4, contiguous shards of {sizes[0]} per GPU")
            conf = {"workload": workload, "batch_total": B_total, "batch_per_gpu": sizes[0], "knots": N,
                    "parallelism": f"batch-shard x{world}" + (" + RCCL gather to rank 0" if world > 1 else ""),
                    "shard_assignment": "contiguous, fixed: rank r solves problems [lo_r, hi_r)"}
        else:
            workload = f"BASELINE.json configs[1]: batch={B}/GPU random SE(3) starts -> hover, {N} knots, fp64, model A, seed 2"
            conf = {"workload": workload, "batch_per_gpu": B, "knots": N,
                    "parallelism": f"batch-shard x{world}" + (" + RCCL gather to rank 0" if world > 1 else ""),
                    "shard_assignment": ("one shard" if world == 1 else
                                         f"every rank solves the {B} problems of the N = 1 line (--shards same, diagnostic)" if same else
                                         f"distinct: {world} shards of {B} distinct problems per step; rank r solves shard (r + step) mod {world}")}
        line = {
            "metric": "iLQR solves/sec (batch, 100-knot SE(3) quadrotor)", "value": value, "unit": "solves/s",
            "value_region": "device-resident: inputs in HBM when the timed region starts, results left in HBM (the task's contract for `value`); "
                            "the metric as SURVEY.md section 8(d) words it, host buffers in -> host buffers out over PCIe, is `host_to_host.value`",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "config": conf,
            "iters_mean": float(iters.mean()), "iters_max": int(iters.max()),
            "status_counts": np.bincount(status, minlength=4).tolist(),
            "knot_steps_per_s": knot_steps / dt,
            "gather_ms": gather_ms,
            "shard_rounds": shard_rounds_list,  # per shard: rollouts of its slowest problem (the straggler sets a shard's time)
            "same_shard": same_shard, "sharded_c_abi": sharded_abi,
            "roofline": roofline, "cpu_baseline": cpu, "host_to_host": h2h, "large_batch": large, "serving": serving,
            "single_solve": single, "reference_faithful": faithful,
            **({"rehearse_nccl": {"backend": dist.get_backend(), "world_size": 1,
                                  "what": "every step's trajectories and costs sent to rank 0 itself through gather_to_root (ncclSend + ncclRecv in one group)"}}
               if rehearse else {}),
        }
    if rehearse and rank == 0:  # the rows that went through RCCL are the rows that were sent
        k = (step_no[0] - 1) & 1
        assert torch.equal(g_traj, out_traj[k]) and torch.equal(g_cost, out_cost[k]), "rehearsal: gathered rows differ from the solver's"
    # ---- the product's own multi-GPU path (never `value`): the same step through qilqr_solve_batch_sharded_device, in a
    # child process of rank 0, started only AFTER every rank has destroyed its process group (torch's RCCL communicators are gone), closed its
    # solver and -- ranks other than 0 -- left: the child's ncclCommInitAll then finds devices nobody else holds a communicator on.  The other
    # ranks report through the rendezvous store (kept alive by rank 0's reference) just before they return; a rank that cannot is waited for
    # by time.  Whatever the leg does, the line is printed.
    store = None
    if world > 1 and not args.no_sharded_c_abi:
        try:
            store = dist.distributed_c10d._get_default_store()
        except Exception:
            store = None
    if world > 1 or rehearse:
        dist.barrier()
        dist.destroy_process_group()
    if world > 1 and not args.no_sharded_c_abi:
        try:
            solver.close()
        except Exception:
            pass
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        if rank != 0:
            try:
                if store is not None:
                    store.add("qilqr_rank_released", 1)
            except Exception:
                pass
            return
        released, t_wait = 0, time.perf_counter()
        while released < world - 1 and time.perf_counter() - t_wait < 20.0:
            try:
                released = int(store.add("qilqr_rank_released", 0)) if store is not None else 0
            except Exception:
                released = 0
            if released < world - 1:
                time.sleep(0.05)
        time.sleep(0.5)  # (a rank reports just before its interpreter exits: let the processes go)
        try:
            line["sharded_c_abi"] = sharded_c_abi_leg(args, ref_cost)
        except Exception as e:  # (json, np.load, a missing file: the leg's failure is a field of the line, never a missing line)
            line["sharded_c_abi"] = {"error": f"{type(e).__name__}: {e}"}
        if isinstance(line["sharded_c_abi"], dict):
            line["sharded_c_abi"]["ranks_released_before_start"] = released
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
