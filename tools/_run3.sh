cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/full_gpu.log 2>&1; tail -3 gpurun_out/full_gpu.log
