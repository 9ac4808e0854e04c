#!/usr/bin/env python3
"""Condense bench.py's JSON line (stdin) to one short line; extra argv are echoed as a label."""
import json
import sys

j = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = j.get("roofline") or {}
print(" ".join(sys.argv[1:]), round(j["value"]), "solves/s", round(j["ms_per_step"], 3), "ms/step",
      r.get("kernel"), round(r.get("avg_launch_us", 0), 1), "us", "frac", round(r.get("frac", 0), 3),
      r.get("warmup_avg_launch_us"), r.get("kernels_ms"))
