#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel from a counter_collection.csv (average per dispatch).
usage: pmc_sum.py <dir> [kernel substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if sub not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
for k in acc:
    n = len(nd[k])
    print(k, "dispatches", n)
    for c, v in sorted(acc[k].items()):
        print("   %-28s %14.1f per dispatch" % (c, v / n))
