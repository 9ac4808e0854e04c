#!/bin/bash
# Diagnostic (round 6): the rollout ordinal from which a batch beyond 4096 takes k_rollout16 (and may change over to k_round), re-swept with the tail on k_round
export PYTHONPATH=. GPU_MAX_HW_QUEUES=8
L=quadrotorilqr_amd/lib
for B in 8192 16384 65536; do
  for v in ship from10 from12 from14 ship from10 from12 from14; do
    lib=$L/libquadrotor_ilqr_$v.so; [ $v = ship ] && lib=$L/libquadrotor_ilqr.so
    QILQR_LIB=$lib python3 profiles/microbench/one_config.py $B reps=7 | sed "s/^/$v   /"
  done
done
