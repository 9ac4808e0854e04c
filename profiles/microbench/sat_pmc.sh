#!/bin/bash
# Round 6: what the kernels of the saturated regime ISSUE.  One launch each of k_linearize and k_backward4 (GFAC form) with EVERY trajectory
# live (pass_time.py: qilqr_backwards_pass at B = 8192) and whole B = 8192 solves (one_config.py: k_rollout3 / k_rollout16 / compaction too);
# per-dispatch SQ counters, one counter set per pass (no trace domains besides --kernel-trace with --pmc).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=. GPU_MAX_HW_QUEUES=8
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU"; do
  i=$((i + 1))
  rm -rf gpurun_out/sp_$i gpurun_out/sq_$i
  # shellcheck disable=SC2086
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/sp_$i -- python3 profiles/microbench/pass_time.py 8192 gfac > gpurun_out/sp_$i.log 2>&1 || echo "pass $i ($set) failed: $(tail -2 gpurun_out/sp_$i.log)"
  # shellcheck disable=SC2086
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/sq_$i -- python3 profiles/microbench/one_config.py 8192 reps=2 > gpurun_out/sq_$i.log 2>&1 || echo "pass $i ($set) failed: $(tail -2 gpurun_out/sq_$i.log)"
done
for i in 1 2 3 4; do echo "== all-live launches, set $i"; python3 profiles/microbench/pmc_sum.py gpurun_out/sp_$i k_; done > gpurun_out/r06_sat_pmc_alllive.txt 2>&1
for i in 1 2 3 4; do echo "== whole B = 8192 solves, set $i"; python3 profiles/microbench/pmc_sum.py gpurun_out/sq_$i k_; done > gpurun_out/r06_sat_pmc_solve.txt 2>&1
cat gpurun_out/r06_sat_pmc_alllive.txt
