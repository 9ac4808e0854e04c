#!/bin/bash
# Diagnostic: the shader clock and power rocm-smi reports while bench.py runs (sampled in the background).
# usage (from the repository root, through gpurun): bash profiles/microbench/clocks.sh
python3 bench.py --steps 1000 --warmup 5 --no-cpu-baseline --no-serving > gpurun_out/clocks_bench.log 2>&1 &
BP=$!
for i in $(seq 1 40); do
  echo "t=$i $(rocm-smi --showclocks 2>/dev/null | grep -i 'sclk' | head -1 | sed 's/.*level//') $(rocm-smi --showpower 2>/dev/null | grep -i 'power (' | head -1 | sed 's/.*://')"
  kill -0 $BP 2>/dev/null || break
done
wait $BP
tail -c 200 gpurun_out/clocks_bench.log
