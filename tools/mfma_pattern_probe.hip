// Dev tool: cycles per v_mfma_f32_32x32x2_f32 (s_memtime, i.e. independent of the DVFS clock) for the issue patterns of the fp32 conv
// K loop: number of independent accumulators, LDS reads / VMEM loads interleaved, operand registers changing, 1 or 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_pattern_probe.hip -o tools/mfma_pattern_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 4 = MODE 2 with the reads pinned between the MFMAs (sched_group_barrier); MODE 5 = MODE 4 with the accumulators in AGPRs (inline asm).
// MODE 0: bare MFMAs, same operands.  1: operands rotate through 8 registers.  2: + one ds_read_b128 per 4*MI MFMAs (used as the next
// operands, like the conv loop).  3: + two global_load_dwordx4 per 8*MI MFMAs (weight fragments).
template <int MI>
__global__ __launch_bounds__(512) void probe_pinned(float* out, const f32x4* w, int iters, unsigned long long* cyc, int agpr) {
    __shared__ f32x4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = f32x4{1e-3f * i, 2e-3f, 3e-3f, 4e-3f};
    __syncthreads();
    f32x16 acc[MI];
    for (int m = 0; m < MI; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    f32x4 xa[MI], xb[MI];
    for (int m = 0; m < MI; ++m) { xa[m] = lds[threadIdx.x + m * 64]; xb[m] = lds[threadIdx.x + 512 + m * 64]; }
    f32x4 wa = w[threadIdx.x], wb = w[threadIdx.x + 512];
    int la = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (!agpr) {
        for (int i = 0; i < iters; ++i) {
            f32x4 na[MI], nb[MI];
#pragma unroll
            for (int m = 0; m < MI; ++m) { na[m] = lds[(la + m * 64) & 4095]; nb[m] = lds[(la + 512 + m * 64) & 4095]; }
            la += 37;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < MI; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s], xa[m][s], acc[m], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < MI; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[s], xb[m][s], acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 2 * MI; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * MI, 0);
#pragma unroll
            for (int m = 0; m < MI; ++m) { xa[m] = na[m]; xb[m] = nb[m]; }
        }
    } else {
        for (int i = 0; i < iters; ++i) {
            f32x4 na[MI], nb[MI];
#pragma unroll
            for (int m = 0; m < MI; ++m) { na[m] = lds[(la + m * 64) & 4095]; nb[m] = lds[(la + 512 + m * 64) & 4095]; }
            la += 37;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < MI; ++m) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(wa[s]), "v"(xa[m][s]));
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < MI; ++m) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(wb[s]), "v"(xb[m][s]));
#pragma unroll
            for (int m = 0; m < MI; ++m) { xa[m] = na[m]; xb[m] = nb[m]; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int m = 0; m < MI; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MI, int MODE>
__global__ __launch_bounds__(512) void probe(float* out, const f32x4* w, int iters, unsigned long long* cyc) {
    __shared__ f32x4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = f32x4{1e-3f * i, 2e-3f, 3e-3f, 4e-3f};
    __syncthreads();
    f32x16 acc[MI];
    for (int m = 0; m < MI; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    f32x4 xa[MI], xb[MI];
    for (int m = 0; m < MI; ++m) { xa[m] = lds[threadIdx.x + m * 64]; xb[m] = lds[threadIdx.x + 512 + m * 64]; }
    f32x4 wa = w[threadIdx.x], wb = w[threadIdx.x + 512];
    const f32x4* wp = w + threadIdx.x;
    int la = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        f32x4 na[MI], nb[MI];
        if (MODE >= 2) {
#pragma unroll
            for (int m = 0; m < MI; ++m) { na[m] = lds[(la + m * 64) & 4095]; }
        }
        f32x4 nwa = wa, nwb = wb;
        if (MODE >= 3) { nwa = wp[0]; nwb = wp[512]; wp += 1024; if ((i & 63) == 63) wp = w + threadIdx.x; }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < MI; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(MODE >= 1 ? wa[s] : wa[0], MODE >= 1 ? xa[m][s] : xa[0][0], acc[m], 0, 0, 0);
        if (MODE >= 2) {
#pragma unroll
            for (int m = 0; m < MI; ++m) { nb[m] = lds[(la + 512 + m * 64) & 4095]; }
            la += 37;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < MI; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(MODE >= 1 ? wb[s] : wa[0], MODE >= 1 ? xb[m][s] : xa[0][0], acc[m], 0, 0, 0);
        if (MODE >= 2) {
#pragma unroll
            for (int m = 0; m < MI; ++m) { xa[m] = na[m]; xb[m] = nb[m]; }
        }
        wa = nwa; wb = nwb;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int m = 0; m < MI; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MI, int MODE>
void run(const char* label, int threads, float* out, f32x4* w, unsigned long long* cyc) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    if (MODE >= 4) probe_pinned<MI><<<256, threads>>>(out, w, 50, cyc, MODE == 5); else
    probe<MI, MODE><<<256, threads>>>(out, w, 50, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    if (MODE >= 4) probe_pinned<MI><<<256, threads>>>(out, w, iters, cyc, MODE == 5); else
    probe<MI, MODE><<<256, threads>>>(out, w, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
    const double waves_per_simd = threads / 256.0;
    const double mfma_per_wave = (double)iters * 8 * MI;
    printf("%-44s MI=%d waves/SIMD=%.0f: %.1f cycles per MFMA per SIMD (64 = peak), %.2f GHz, %.1f TFLOP/s\n", label, MI, waves_per_simd,
           avg / (mfma_per_wave * waves_per_simd), avg / (ms * 1e6), 256.0 * (threads / 64) * mfma_per_wave * 2 * 32 * 32 * 2 / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    f32x4* w; hipMalloc(&w, 1 << 22);
    {
        float* hw = (float*)malloc(1 << 22);
        unsigned st = 1;
        for (int i = 0; i < (1 << 20); ++i) { st = st * 1664525u + 1013904223u; hw[i] = ((st >> 8) & 0xffff) / 32768.f - 1.f; }
        hipMemcpy(w, hw, 1 << 22, hipMemcpyHostToDevice);
    }
    unsigned long long* cyc; hipMalloc(&cyc, 256 * 8);
    for (int threads : {256, 512}) {
        run<4, 0>("bare, same operands", threads, out, w, cyc);
        run<2, 0>("bare, same operands", threads, out, w, cyc);
        run<1, 0>("bare, same operands", threads, out, w, cyc);
        run<4, 1>("operands rotate", threads, out, w, cyc);
        run<2, 1>("operands rotate", threads, out, w, cyc);
        run<4, 2>("+ ds_read_b128 per 4*MI", threads, out, w, cyc);
        run<2, 2>("+ ds_read_b128 per 4*MI", threads, out, w, cyc);
        run<4, 4>("ds_read per 4*MI, pinned between MFMAs", threads, out, w, cyc);
        run<2, 4>("ds_read per 4*MI, pinned between MFMAs", threads, out, w, cyc);
        run<4, 5>("ds_read per 4*MI, accumulators in AGPRs (asm)", threads, out, w, cyc);
        run<2, 5>("ds_read per 4*MI, accumulators in AGPRs (asm)", threads, out, w, cyc);
        run<4, 3>("+ ds_read + 2 global loads per 8*MI", threads, out, w, cyc);
        run<2, 3>("+ ds_read + 2 global loads per 8*MI", threads, out, w, cyc);
    }
    return 0;
}
