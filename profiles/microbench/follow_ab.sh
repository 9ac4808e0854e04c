#!/bin/bash
# Diagnostic (round 6): k_round with the candidates' linearisation behind the rollout (round_follow) against the linearisation after it
# (-DQILQR_ROUND_NO_FOLLOW) and against the identity assignment of roles to wavefronts (-DQILQR_ROUND_IDENTITY_ROLES): three builds, one
# configuration per process, alternately.
export PYTHONPATH=. GPU_MAX_HW_QUEUES=8
L=quadrotorilqr_amd/lib
for rep in 1 2; do
  for v in nofollow idroles ship; do
    lib=$L/libquadrotor_ilqr_$v.so; [ $v = ship ] && lib=$L/libquadrotor_ilqr.so
    echo "== $v"
    QILQR_LIB=$lib python3 profiles/microbench/one_config.py 1024 seed=2 reps=30
    QILQR_LIB=$lib python3 profiles/microbench/one_config.py 256 seed=2 reps=30
    QILQR_LIB=$lib python3 profiles/microbench/one_config.py 1 seed=2 reps=30
  done
done
