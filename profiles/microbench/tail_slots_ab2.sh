#!/bin/bash
# Diagnostic (round 6): the changeover to k_round at a fixed number of slots per sub-batch (1024, 1280) against 3 blocks per CU over the sub-batches
export PYTHONPATH=. GPU_MAX_HW_QUEUES=8
L=quadrotorilqr_amd/lib
for B in 4096 8192 16384 65536; do
  for v in tail3 pp1024 pp1280 tail3 pp1024 pp1280; do
    QILQR_LIB=$L/libquadrotor_ilqr_$v.so python3 profiles/microbench/one_config.py $B reps=7 | sed "s/^/$v   /"
  done
done
