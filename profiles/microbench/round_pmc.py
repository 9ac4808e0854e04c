#!/usr/bin/env python3
"""Reads the per-dispatch counter files of round_pmc.sh: for every k_round dispatch of the LAST solve of the probe (one round per launch),
its duration and counters; prints the rounds in order and the means over the rounds with every trajectory running (rounds 1..6) and over
the late rounds (at most one running trajectory per block: rounds 22..30).
usage: python3 profiles/microbench/round_pmc.py <prefix> <passes>"""
import csv
import glob
import sys
from collections import defaultdict

prefix, passes = sys.argv[1], int(sys.argv[2])
rows = {}  # dispatch ordinal among k_round launches -> {name: value}
for p in range(1, passes + 1):
    files = glob.glob(f"{prefix}{p}/**/*counter_collection.csv", recursive=True)
    traces = glob.glob(f"{prefix}{p}/**/*kernel_trace.csv", recursive=True)
    if not files:
        print(f"pass {p}: no counter file")
        continue
    dur = {}
    for tf in traces:
        for r in csv.DictReader(open(tf)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    per = defaultdict(dict)
    order = []
    for r in csv.DictReader(open(files[0])):
        if "k_round" not in r["Kernel_Name"]:
            continue
        d = r["Dispatch_Id"]
        if d not in per:
            order.append(d)
        per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if d in dur:
            per[d]["us"] = dur[d]
    # the probe solves three times; rounds per solve = len(order) / 3
    n = len(order) // 3
    for k, d in enumerate(order[2 * n:]):
        rows.setdefault(k, {}).update({(f"us_pass{p}" if c == "us" else c): v for c, v in per[d].items()})
if not rows:
    sys.exit("no k_round dispatches found")
names = sorted({c for r in rows.values() for c in r})
print("round  " + "  ".join(f"{c:>26s}" for c in names))
for k in sorted(rows):
    print(f"{k + 1:5d}  " + "  ".join(f"{rows[k].get(c, float('nan')):26.1f}" for c in names))


def mean(lo, hi, c):
    v = [rows[k][c] for k in rows if lo <= k + 1 <= hi and c in rows[k]]
    return sum(v) / len(v) if v else float("nan")


print("\nmeans                              rounds 1..6 (all running)   rounds 22..30 (<= 1 per block)   ratio")
for c in names:
    a, b = mean(1, 6, c), mean(22, 30, c)
    print(f"{c:>32s}  {a:26.1f}  {b:30.1f}  {a / b if b else float('nan'):6.3f}")
us = [c for c in names if c.startswith("us_pass")]
if "GRBM_GUI_ACTIVE" in names and us:
    for lo, hi, lab in ((1, 6, "all running"), (22, 30, "<= 1 per block")):
        print(f"shader clock, {lab}: {mean(lo, hi, 'GRBM_GUI_ACTIVE') / mean(lo, hi, 'us_pass1') / 1e3:.3f} GHz (GRBM_GUI_ACTIVE / duration; per-XCD sums divide by 8 if > 10)")
