#!/bin/bash
# Round 5: why does a round of k_round take ~150 us with four running trajectories per block and ~127 us with one?
# One k_round launch per round (rounds_per_launch = 1, angle_probe.py: configs[1]), per-dispatch counters; the first rounds of a solve
# have every trajectory running, the late ones at most one per block.  GRBM_GUI_ACTIVE / dispatch duration = the shader clock the
# round ran at; the SQ counters split the cycles.  One counter set per pass (no trace domains besides --kernel-trace with --pmc).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=. GPU_MAX_HW_QUEUES=8
rocprofv3 -L > gpurun_out/r05_counters_available.txt 2>&1
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
  i=$((i + 1))
  rm -rf gpurun_out/rp_$i
  # shellcheck disable=SC2086
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/rp_$i -- python3 profiles/microbench/angle_probe.py 0.785 > gpurun_out/rp_$i.log 2>&1 || echo "pass $i ($set) failed: $(tail -2 gpurun_out/rp_$i.log)"
done
python3 profiles/microbench/round_pmc.py gpurun_out/rp_ 5 > gpurun_out/r05_round_pmc.txt 2>&1
cat gpurun_out/r05_round_pmc.txt
