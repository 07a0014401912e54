#!/bin/bash
# Round-end evidence on the GPU box (through gpurun, from the repo root): the full GPU suite, every rocprofv3 pass of tools/collect_profiles.sh,
# then the default bench line.   tools/run_round_end.sh <tag>      Output: gpurun_out/
tag=${1:-r04}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/${tag}_gpu_tests.log 2>&1; tail -2 gpurun_out/${tag}_gpu_tests.log
bash tools/collect_profiles.sh $tag > gpurun_out/collect_${tag}.log 2>&1
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 400 gpurun_out/${tag}_bench.json
