#!/bin/bash
# Diagnostic (round 6): the late rounds of a batch beyond 4096 on the combined launch / k_round (QILQR_LATE_TAIL) against three launches per round
# to the end: two builds, one configuration per process, alternately.  usage: bash profiles/microbench/tail_ab.sh
#   other build: hipcc ... -DQILQR_LATE_TAIL=0 -o quadrotorilqr_amd/lib/libquadrotor_ilqr_notail.so quadrotorilqr_amd/csrc/ilqr_capi.hip
export PYTHONPATH=. GPU_MAX_HW_QUEUES=8
for B in 4096 8192 16384 65536; do
  for rep in 1 2; do
    QILQR_LIB=quadrotorilqr_amd/lib/libquadrotor_ilqr_notail.so python3 profiles/microbench/one_config.py $B reps=7 | sed 's/^/three launches  /'
    python3 profiles/microbench/one_config.py $B reps=7 | sed 's/^/late tail       /'
  done
done
