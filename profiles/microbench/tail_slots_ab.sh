#!/bin/bash
# Diagnostic (round 6): the changeover of a compacted batch to the combined launch / k_round at 1 (ships), 2, 3 blocks of four per CU.
export PYTHONPATH=. GPU_MAX_HW_QUEUES=8
L=quadrotorilqr_amd/lib
for B in 2048 4096 8192 16384 65536; do
  for v in ship tail2 tail3 ship tail2 tail3; do
    lib=$L/libquadrotor_ilqr_$v.so; [ $v = ship ] && lib=$L/libquadrotor_ilqr.so
    QILQR_LIB=$lib python3 profiles/microbench/one_config.py $B reps=7 | sed "s/^/$v   /"
  done
done
