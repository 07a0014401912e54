cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 300 tools/bf16x3_timeline.bin gemm > gpurun_out/r06c_gemm_timeline.txt 2>&1
cat gpurun_out/r06c_gemm_timeline.txt | cut -c1-330
