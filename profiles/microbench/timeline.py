#!/usr/bin/env python3
"""Diagnostic: where one batch solve's wall time goes, from a rocprofv3 --kernel-trace CSV of bench.py.
Per solve (delimited by k_begin): span first start -> last end, per-kernel sums, idle time, and the cost of
every round (k_backward start -> next k_backward start) so the tail rounds can be compared with the full ones.
usage: timeline.py <dir containing *_kernel_trace.csv>"""
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
            n = r["Kernel_Name"]
            if "qilqr" not in n:
                continue
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].split("<")[0].split("::")[-1]))
rows.sort()
solves, cur = [], []
for r in rows:
    if r[2] == "k_begin" and cur:
        solves.append(cur)
        cur = []
    cur.append(r)
solves.append(cur)
for si, sv in enumerate(solves[-int(os.environ.get('TIMELINE_SOLVES', '3')):]):
    t0, t1 = sv[0][0], max(r[1] for r in sv)
    busy = defaultdict(float)
    cnt = defaultdict(int)
    for s, e, n in sv:
        busy[n] += (e - s) / 1e3
        cnt[n] += 1
    idle = sum(max(0, b[0] - a[1]) for a, b in zip(sv, sv[1:])) / 1e3
    print(f"solve {si}: span {(t1 - t0) / 1e3:.1f} us, kernels {len(sv)}, idle between kernels {idle:.1f} us")
    for n in sorted(busy, key=lambda k: -busy[k]):
        print(f"    {n:16s} n={cnt[n]:4d} sum={busy[n]:8.1f} us mean={busy[n] / cnt[n]:7.2f}")
    bw = [i for i, r in enumerate(sv) if r[2].startswith("k_backward") or r[2] == "k_round"]
    print(f"    before first k_backward: {(sv[bw[0]][0] - t0) / 1e3:.1f} us; after last k_backward start: {(t1 - sv[bw[-1]][0]) / 1e3:.1f} us")
    per = []
    for a, b in zip(bw, bw[1:]):
        seg = sv[a:b]
        per.append(((sv[b][0] - sv[a][0]) / 1e3, [round((e - s) / 1e3, 1) for s, e, _ in seg]))
    for k, (d, parts) in enumerate(per):
        print(f"    round {k:2d}: {d:7.1f} us  {parts}")
