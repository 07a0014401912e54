// Dev tool: what does a 16-byte-per-lane load (ds_read_b128 / global_load_dwordx4) issued next to v_mfma_f32_32x32x2_f32 cost the matrix pipe?
// The loads' results are NOT operands of the MFMAs (they are summed into a side register after the loop section), so waits on them
// are off the MFMA path: any slowdown is a shared-resource effect.  R = loads per block of 8 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int R, int KIND>  // KIND 0: ds_read_b128, 1: global_load_dwordx4, 2: v_mov (VALU writes of 4 registers)
__global__ __launch_bounds__(512) void probe(float* out, const f32x4* w, int iters) {
    __shared__ f32x4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = f32x4{1e-3f * i, 2e-3f, 3e-3f, 4e-3f};
    __syncthreads();
    f32x16 acc[2];
    for (int m = 0; m < 2; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    float a[4], b[4];
    for (int s = 0; s < 4; ++s) { a[s] = 1e-3f * (threadIdx.x + s); b[s] = 2e-3f * (threadIdx.x - s); }
    f32x4 side[R > 0 ? R : 1];
    for (int r = 0; r < (R > 0 ? R : 1); ++r) side[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    int la = threadIdx.x;
    const f32x4* wp = w + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        f32x4 ld[R > 0 ? R : 1];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (KIND == 0) ld[r] = lds[(la + 64 * r) & 4095];
            else if (KIND == 1) ld[r] = wp[512 * r];
            else { f32x4 t = side[r]; asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "=v"(t[0]), "=v"(t[1]), "=v"(t[2]), "=v"(t[3]) : "v"(a[0])); ld[r] = t; }
        }
        la += 37;
        wp += 512 * (R > 0 ? R : 1);
        if ((i & 31) == 31) wp = w + threadIdx.x;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[(s + m) & 3], acc[m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; ++r) side[r] += ld[r];  // consumed after the MFMA block: the wait is behind 8 MFMAs
    }
    float s = 0.f;
    for (int m = 0; m < 2; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    for (int r = 0; r < (R > 0 ? R : 1); ++r) s += side[r][0] + side[r][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int R, int KIND>
void run(int threads, float* out, f32x4* w) {
    const int iters = 8000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<R, KIND><<<256, threads>>>(out, w, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<R, KIND><<<256, threads>>>(out, w, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 8 * (threads / 256);
    // assume the clock this chip sustains here is unknown: report ns per MFMA and the implied cycles at 2.35 GHz
    const double ns = ms * 1e6 / mfma_per_simd;
    printf("%s x%d per 8 MFMAs, %d wave(s)/SIMD: %.2f ns per MFMA (27.2 = 64 cycles at 2.35 GHz)  %.1f TFLOP/s\n",
           KIND == 0 ? "ds_read_b128      " : KIND == 1 ? "global_load_dwordx4" : "4 x v_mov_b32      ", R, threads / 256, ns, 256.0 * 4 * mfma_per_simd * 4096 / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    f32x4* w; hipMalloc(&w, 1 << 24);
    hipMemset(w, 0, 1 << 24);
    for (int threads : {256, 512}) {
        run<0, 0>(threads, out, w);
        run<1, 0>(threads, out, w);
        run<2, 0>(threads, out, w);
        run<4, 0>(threads, out, w);
        run<8, 0>(threads, out, w);
        run<1, 1>(threads, out, w);
        run<2, 1>(threads, out, w);
        run<4, 1>(threads, out, w);
        run<2, 2>(threads, out, w);
        run<8, 2>(threads, out, w);
    }
    return 0;
}
