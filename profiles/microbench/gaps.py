#!/usr/bin/env python3
"""Diagnostic: idle time between consecutive kernels on the stream, from a rocprofv3 --kernel-trace CSV.
usage: gaps.py <dir containing *_kernel_trace.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict

paths = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()


def short(n):
    for k in ("k_backward", "k_rollout3", "k_rollout2", "k_rollout", "k_linearize", "k_init", "k_gather", "k_retile",
              "k_seed_search", "k_accept"):
        if k in n:
            return k
    return n[:30]


gap = defaultdict(list)
dur = defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    dur[short(n0)].append(e0 - s0)
    if s1 - e0 < 200000:  # ignore the pauses between solves
        gap[short(n0) + " -> " + short(n1)].append(s1 - e0)
print("durations (us):")
for k, v in sorted(dur.items()):
    print(f"  {k:14s} n={len(v):6d} mean={sum(v) / len(v) / 1e3:8.2f}")
print("gaps (us):")
for k, v in sorted(gap.items()):
    v = sorted(v)
    print(f"  {k:28s} n={len(v):6d} mean={sum(v) / len(v) / 1e3:7.2f} median={v[len(v) // 2] / 1e3:7.2f}")
