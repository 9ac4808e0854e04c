cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PYTHONPATH=. GPU_MAX_HW_QUEUES=8
for a in 0.785 0.2; do
  rm -rf gpurun_out/ap_$a
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ap_$a -- python3 profiles/microbench/angle_probe.py $a > gpurun_out/ap_$a.log 2>&1
  grep "^ang" gpurun_out/ap_$a.log
  TIMELINE_SOLVES=1 python3 profiles/microbench/timeline.py gpurun_out/ap_$a | sed -n 1,22p
done
