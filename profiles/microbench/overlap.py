#!/usr/bin/env python3
"""Diagnostic: how much kernels of different streams overlap in time (rocprofv3 --kernel-trace CSV).
usage: overlap.py <dir>"""
import csv
import glob
import os
import sys

rows = []
for p in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-28:],
                         r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t0 = rows[0][0]
busy = 0
union_end = 0
total = 0
for s, e, *_ in rows:
    total += e - s
    if s > union_end:
        busy += e - s
        union_end = e
    elif e > union_end:
        busy += e - union_end
        union_end = e
print("kernels", len(rows), "sum of durations %.2f ms" % (total / 1e6), "union (wall busy) %.2f ms" % (busy / 1e6),
      "overlap factor %.2f" % (total / max(busy, 1)))
print("queues:", sorted(set(r[3] for r in rows)), "streams:", sorted(set(r[4] for r in rows))[:12])
mid = len(rows) // 2
for s, e, n, q, st in rows[mid:mid + 16]:
    print(f"  {(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:7.1f}  q{q} s{st} {n}")
