// Dev tool: sustained rate of v_mfma_f32_32x32x2_f32 alone (the ceiling of the exact-fp32 conv path).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MI>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    f32x16 acc[MI];
    for (int m = 0; m < MI; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int m = 0; m < MI; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
    }
    float s = 0.f;
    for (int m = 0; m < MI; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {256, 512, 1024}) {
        const int iters = 20000;
        probe<4><<<wgs, 256>>>(d, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        probe<4><<<wgs, 256>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)wgs * 4 * iters * 8 * 4 * 2.0 * 32 * 32 * 2;
        printf("wgs %d: %.2f ms  %.1f TFLOP/s\n", wgs, ms, flops / ms / 1e9);
    }
    return 0;
}
