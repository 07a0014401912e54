// Micro-benchmark of the bf16x3 conv inner loop (ablation probe, not part of the product).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// V bit0: read A from LDS each step; bit1: stream B from global; bit2: barrier per "chunk"
template <int V, int MI, int NJ, int ORD = 0>
__global__ __launch_bounds__(256) void probe(const bf16x8* __restrict__ w, float* out, int ntaps, int nchunks) {
    constexpr int NC16 = 4, CH = 64, PITCH = CH * 4 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, g = lane >> 5;
    for (int i = tid; i < 178 * PITCH / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x16 acc[MI][NJ];
    for (int mi = 0; mi < MI; ++mi) for (int j = 0; j < NJ; ++j) for (int r = 0; r < 16; ++r) acc[mi][j][r] = 0.f;
    const bf16x8* wp = w + (size_t)(blockIdx.x % 8) * 4096 + wave * 1024 + lane;
    bf16x8 bq[NC16][NJ][2];
    for (int u = 0; u < NC16; ++u) for (int j = 0; j < NJ; ++j) { bq[u][j][0] = wp[(u * NJ + j) * 128]; bq[u][j][1] = wp[(u * NJ + j) * 128 + 64]; }
    bf16x8 ah[MI], al[MI];
    for (int mi = 0; mi < MI; ++mi) { ah[mi] = *reinterpret_cast<const bf16x8*>(smem + (li + mi * 32) * PITCH + g * 16); al[mi] = ah[mi]; }
    for (int c = 0; c < nchunks; ++c) {
        for (int t = 0; t < ntaps; ++t) {
            const char* arow = smem + (li + t * 5) * PITCH + g * 16;
#pragma unroll
            for (int u = 0; u < NC16; ++u) {
                bf16x8 bh[NJ], bl[NJ];
                for (int j = 0; j < NJ; ++j) { bh[j] = bq[u][j][0]; bl[j] = bq[u][j][1]; }
                if (V & 2) for (int j = 0; j < NJ; ++j) { bq[u][j][0] = wp[(u * NJ + j) * 128]; bq[u][j][1] = wp[(u * NJ + j) * 128 + 64]; }
                if (V & 1) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        ah[mi] = *reinterpret_cast<const bf16x8*>(arow + mi * 32 * PITCH + u * 32);
                        al[mi] = *reinterpret_cast<const bf16x8*>(arow + mi * 32 * PITCH + CH * 2 + u * 32);
                    }
                }
                if (ORD == 0) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[mi], acc[mi][j], 0, 0, 0);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[mi], acc[mi][j], 0, 0, 0);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[mi], acc[mi][j], 0, 0, 0);
                }
                } else {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[mi], acc[mi][j], 0, 0, 0);
                        acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[mi], acc[mi][j], 0, 0, 0);
                        acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[mi], acc[mi][j], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (V & 2) wp += NC16 * NJ * 128;
        }
        if (V & 4) __syncthreads();
        if (V & 2) wp -= ntaps * NC16 * NJ * 128;
    }
    float s = 0.f;
    for (int mi = 0; mi < MI; ++mi) for (int j = 0; j < NJ; ++j) for (int r = 0; r < 16; ++r) s += acc[mi][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int V, int MI, int NJ, int ORD = 0>
void run(const char* name, const bf16x8* w, float* out, int grid, int lds_kb) {
    const int ntaps = 11, nchunks = 16;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<V, MI, NJ, ORD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((probe<V, MI, NJ, ORD>), dim3(grid), dim3(256), lds_kb * 1024, 0, w, out, ntaps, nchunks);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = 5.0 * grid * 4 * (double)nchunks * ntaps * 4 * 3 * MI * NJ * 32768.0;
    printf("%-34s MI=%d NJ=%d grid=%d lds=%dKB  %.3f ms  %.0f TF (bf16 MFMA)  %.0f TF-alg\n", name, MI, NJ, grid, lds_kb, ms / 5, mf / (ms * 1e-3) / 1e12, mf / 3 / (ms * 1e-3) / 1e12);
}

int main() {
    bf16x8* w; float* out;
    hipMalloc(&w, 64 << 20); hipMemset(w, 0x11, 64 << 20); hipMalloc(&out, 4096 * 256 * 4);
    run<0, 4, 1>("mfma only term-major", w, out, 256, 96);
    run<0, 4, 1, 1>("mfma only chained", w, out, 256, 96);
    run<3, 4, 1>("A+B term-major 1WG/CU", w, out, 256, 96);
    run<3, 4, 1, 1>("A+B chained 1WG/CU", w, out, 256, 96);
    run<0, 4, 2>("mfma only term-major", w, out, 256, 96);
    run<0, 4, 2, 1>("mfma only chained", w, out, 256, 96);
    run<0, 4, 1>("mfma only", w, out, 512, 48);
    run<1, 4, 1>("+ A from LDS", w, out, 512, 48);
    run<2, 4, 1>("+ B from global", w, out, 512, 48);
    run<3, 4, 1>("A + B", w, out, 512, 48);
    run<7, 4, 1>("A + B + barrier/chunk", w, out, 512, 48);
    run<3, 4, 1>("A + B, 1 WG/CU", w, out, 256, 96);
    run<3, 4, 1>("A + B, 3 WG/CU", w, out, 768, 48);
    run<0, 4, 2>("mfma only", w, out, 512, 48);
    run<1, 4, 2>("+ A from LDS", w, out, 512, 48);
    run<3, 4, 2>("A + B", w, out, 512, 48);
    run<3, 4, 2>("A + B, 1 WG/CU", w, out, 256, 96);
    run<3, 2, 2>("A + B", w, out, 512, 48);
    run<3, 2, 2>("A + B, 3WG/CU", w, out, 768, 48);
    run<3, 2, 4>("A + B", w, out, 512, 48);
    return 0;
}
