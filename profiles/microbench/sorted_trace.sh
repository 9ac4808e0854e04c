cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PYTHONPATH=. GPU_MAX_HW_QUEUES=8
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tls -- python3 profiles/microbench/sorted_trace.py ${1:-8192} ${2:-1} > gpurun_out/tls.log 2>&1
python3 profiles/microbench/timeline.py gpurun_out/tls > gpurun_out/timeline_sorted.txt
grep -n "solve\|n=" gpurun_out/timeline_sorted.txt
