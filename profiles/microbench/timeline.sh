cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serving --no-profile > gpurun_out/tl.log 2>&1
python3 profiles/microbench/timeline.py gpurun_out/tl > gpurun_out/timeline.txt
tail -60 gpurun_out/timeline.txt
