# Diagnostic (round 4): whole device-resident solves at large batch sizes, library builds in processes of their own on one box
# usage: bash profiles/microbench/compaction_sweep.sh "<B ...>" <variant suffixes, "" = product ...>
export PYTHONPATH=. GPU_MAX_HW_QUEUES=8
Bs=${1:-8192}; shift
for rep in 1 2; do
  for v in "$@"; do
    echo "lib${v:+_}$v"
    QILQR_LIB=$PWD/quadrotorilqr_amd/lib/libquadrotor_ilqr${v:+_}$v.so python3 profiles/microbench/sorted_batch.py $Bs 2>&1 | grep -v amdgpu.ids
  done
done
