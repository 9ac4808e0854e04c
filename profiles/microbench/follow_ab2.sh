#!/bin/bash
# Diagnostic (round 6): with the linearisation behind the rollout, is k_round the better round also where the blocks are full and the chip is
# not (B = 64 ... 512, first rounds: until now k_backward_rollout + k_linearize there)?  -DQILQR_ROUND_ALWAYS against the shipped rule.
export PYTHONPATH=. GPU_MAX_HW_QUEUES=8
L=quadrotorilqr_amd/lib
for B in 64 128 256 512 768; do
  for v in ship always ship always; do
    lib=$L/libquadrotor_ilqr_$v.so; [ $v = ship ] && lib=$L/libquadrotor_ilqr.so
    QILQR_LIB=$lib python3 profiles/microbench/one_config.py $B seed=2 reps=30 | sed "s/^/$v   /"
  done
done
