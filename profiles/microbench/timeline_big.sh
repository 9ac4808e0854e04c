# Diagnostic: per-round kernel timeline of one B = 8192, N = 100 solve (the shard one GPU solves in configs[3]).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PYTHONPATH=. GPU_MAX_HW_QUEUES=8
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlb -- python3 profiles/run_configs.py config4shard > gpurun_out/tlb.log 2>&1
python3 profiles/microbench/timeline.py gpurun_out/tlb > gpurun_out/timeline_big.txt
tail -70 gpurun_out/timeline_big.txt
