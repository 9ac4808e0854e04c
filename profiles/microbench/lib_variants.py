"""Time bench.py's workload with alternative builds of the library (diagnostic only).
usage: python profiles/microbench/lib_variants.py lib1.so lib2.so ...   (each run in a child process)"""
import json
import os
import subprocess
import sys

CHILD = r"""
import sys, json
sys.path.insert(0, %r)
from quadrotorilqr_amd import capi
capi.LIB_PATH = %r
sys.argv = ['bench.py', '--no-cpu-baseline', '--profile-all', '--steps', '10', '--batch', %r]
import runpy
runpy.run_path(%r, run_name='__main__')
"""

if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    batches = os.environ.get("BATCHES", "1024").split(",")
    for lib in sys.argv[1:]:
        for B in batches:
            code = CHILD % (root, os.path.abspath(lib), B, os.path.join(root, "bench.py"))
            out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(lib, B, "FAILED", out.stderr[-400:])
                continue
            j = json.loads(line[-1])
            print(os.path.basename(lib), "B", B, "value", round(j["value"]), "ms", round(j["ms_per_step"], 3),
                  json.dumps(j["roofline"]["kernels_ms"]))
