#!/usr/bin/env python3
"""Condenses a rocprofv3 output tree (profiles/run_rocprof.sh) into small text/JSON summaries that
are committed under profiles/: per-kernel time statistics and per-launch HBM traffic from the PMC
passes (FETCH_SIZE is doubled as MI355X_MICROARCH.md section HBM prescribes for gfx950; both counters
are reported by rocprofv3 in KiB)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def find(root, suffix):
    return sorted(glob.glob(os.path.join(root, "**", "*" + suffix), recursive=True))


def short(name):
    """kernel class: the variants of one pass (k_backward / k_backward2 / k_backward4, k_rollout / k_rollout3 / k_rollout16) share a
    row -- one run uses one variant of each (run_rocprof.sh switches off bench.py's legs at other batch sizes); the
    variants seen are listed beside the row"""
    for k in ("k_solve4", "k_round", "k_backward_rollout", "k_backward", "k_rollout", "k_linearize", "k_compact_plan", "k_compact_move", "k_publish_active",
              "k_debug_capture", "k_accept", "k_init", "k_gather", "k_retile", "k_begin", "k_seed_search"):
        if k in name:
            return k
    return name[:60]


def exact(name):
    m = re.search(r"qilqr::(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def main():
    out, tag = sys.argv[1], sys.argv[2]
    summary = {"tag": tag}
    # ---- kernel trace
    rows = []
    for f in find(os.path.join(out, "trace"), "kernel_trace.csv"):
        rows += list(csv.DictReader(open(f)))
    stat = defaultdict(list)
    seen = defaultdict(set)
    for r in rows:
        stat[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        seen[short(r["Kernel_Name"])].add(exact(r["Kernel_Name"]))
    total = sum(sum(v) for v in stat.values()) or 1.0
    ks = {}
    lines = ["kernel                 calls   total_us    avg_us    min_us    max_us   pct"]
    for k, v in sorted(stat.items(), key=lambda kv: -sum(kv[1])):
        ks[k] = dict(calls=len(v), total_us=sum(v), avg_us=sum(v) / len(v), min_us=min(v), max_us=max(v),
                     pct=100 * sum(v) / total, kernels=sorted(seen[k]))
        lines.append(f"{k:20s} {len(v):7d} {sum(v):10.1f} {sum(v)/len(v):9.2f} {min(v):9.2f} {max(v):9.2f} {100*sum(v)/total:5.1f}   {' '.join(sorted(seen[k]))}")
    summary["kernel_stats"] = ks
    # ---- PMC passes
    for name, sub, mult in (("FETCH_SIZE", "pmc_fetch", 2.0), ("WRITE_SIZE", "pmc_write", 1.0)):
        acc = defaultdict(lambda: [0.0, 0])
        for f in find(os.path.join(out, sub), "counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == name:
                    a = acc[short(r["Kernel_Name"])]
                    a[0] += float(r["Counter_Value"])
                    a[1] += 1
        summary[name] = {k: dict(launches=n, kib_per_launch_raw=v / max(n, 1),
                                 bytes_per_launch_corrected=mult * 1024.0 * v / max(n, 1))
                         for k, (v, n) in acc.items()}
        lines.append("")
        lines.append(f"{name} (KiB raw; corrected bytes = raw x 1024 x {mult:g})")
        for k, d in sorted(summary[name].items()):
            lines.append(f"  {k:20s} launches {d['launches']:6d}  raw KiB/launch {d['kib_per_launch_raw']:12.1f}  "
                         f"corrected MB/launch {d['bytes_per_launch_corrected']/1e6:10.3f}")
    # ---- per-kernel HBM rate: (fetched + written bytes per launch, from the two PMC passes) / average launch duration of the trace
    if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
        lines.append("")
        lines.append("HBM rate per kernel = (FETCH x 2 + WRITE bytes per launch) / avg launch duration of the kernel trace; peak 8000 GB/s")
        summary["hbm_rate"] = {}
        for k, d in ks.items():
            f_, w_ = summary["FETCH_SIZE"].get(k), summary["WRITE_SIZE"].get(k)
            if not f_ or not w_ or d["avg_us"] <= 0:
                continue
            byts = f_["bytes_per_launch_corrected"] + w_["bytes_per_launch_corrected"]
            gbs = byts / (d["avg_us"] * 1e-6) / 1e9
            summary["hbm_rate"][k] = dict(bytes_per_launch=byts, gb_per_s=gbs, frac_of_8tb=gbs / 8000.0)
            lines.append(f"  {k:20s} {byts/1e6:10.3f} MB/launch  {d['avg_us']:9.2f} us  {gbs:9.1f} GB/s  {100*gbs/8000.0:5.1f} % of peak")
    for f in ("bench_trace.log",):
        p = os.path.join(out, f)
        if os.path.exists(p):
            js = [l for l in open(p) if l.startswith("{")]
            if js:
                summary["bench_under_trace"] = json.loads(js[-1])
    here = os.path.dirname(os.path.abspath(__file__))
    open(os.path.join(here, f"{tag}_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")
    json.dump(summary, open(os.path.join(here, f"{tag}_rocprof_summary.json"), "w"), indent=1)
    # scratch copy too (gpurun merges gpurun_out back)
    os.makedirs(os.path.join(out, "summary"), exist_ok=True)
    open(os.path.join(out, "summary", f"{tag}_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")
    json.dump(summary, open(os.path.join(out, "summary", f"{tag}_rocprof_summary.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
