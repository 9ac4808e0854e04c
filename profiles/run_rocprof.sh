#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run from the repo root through gpurun):
#   1. kernel trace + stats of the default bench command
#   2. HBM read and write bytes of the same command, one PMC pass each (MI355X_MICROARCH.md: FETCH_SIZE
#      and WRITE_SIZE do not fit one pass; no trace domains besides --kernel-trace with --pmc)
# Usage: profiles/run_rocprof.sh <tag> [bench args...]
#        QILQR_PROF_CMD="profiles/run_configs.py config3" profiles/run_rocprof.sh <tag>     (another workload: the three passes
#        run `python3 $QILQR_PROF_CMD` instead of bench.py -- e.g. BASELINE.json configs[2])
set -u
# The three runs profile configs[1] alone: bench.py's extra legs (several batches in flight, host buffers, B = 8192) lie
# outside its timed region, run other batch sizes through other kernel variants, and would blur every per-kernel row.
TAG=${1:-r01}; shift || true
set -- --no-serving --no-host-to-host --no-large-batch --no-single-solve --no-reference-faithful "$@"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
if [ -n "${QILQR_PROF_CMD:-}" ]; then
  # shellcheck disable=SC2086
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $QILQR_PROF_CMD > "$OUT/bench_trace.log" 2>&1 &&
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 $QILQR_PROF_CMD > "$OUT/bench_fetch.log" 2>&1 &&
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 $QILQR_PROF_CMD > "$OUT/bench_write.log" 2>&1
else
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench_trace.log" 2>&1 &&
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench_fetch.log" 2>&1 &&
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench_write.log" 2>&1
fi
python3 profiles/summarize.py "$OUT" "$TAG"
