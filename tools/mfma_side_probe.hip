// Dev tool: what does a 16-byte-per-lane load (ds_read_b128 / global_load_dwordx4) issued next to v_mfma_f32_32x32x2_f32 cost the matrix pipe?
// The loads' results are NOT operands of the MFMAs (they are summed into a side register after the loop section), so waits on them
// are off the MFMA path: any slowdown is a shared-resource effect.  R = loads per block of 8 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// KIND 0: ds_read_b128, 1: global_load_dwordx4, 2: 4 x v_mov_b32, 3: 2 x v_pk_add_f32 (4 registers written by 2 instructions),
// 4: 4 x v_add_f32, 5: global_store_dwordx4 (no VGPR written), 6: 2 x ds_read_b64, 7: ds_write_b128 (no VGPR written), 8: 4 x v_fmac_f32 (in place)
template <int R, int KIND>
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
            else if (KIND == 2) { f32x4 t = side[r]; asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "=v"(t[0]), "=v"(t[1]), "=v"(t[2]), "=v"(t[3]) : "v"(a[0])); ld[r] = t; }
            else if (KIND == 3) { f32x4 t = side[r]; asm volatile("v_pk_add_f32 %0, %2, %2\n v_pk_add_f32 %1, %3, %3" : "=v"(*(reinterpret_cast<double*>(&t))), "=v"(*(reinterpret_cast<double*>(&t) + 1)) : "v"(*(reinterpret_cast<double*>(&side[r]))), "v"(*(reinterpret_cast<double*>(&side[r]) + 1))); ld[r] = t; }
            else if (KIND == 4) { f32x4 t; asm volatile("v_add_f32 %0, %4, %4\n v_add_f32 %1, %4, %4\n v_add_f32 %2, %4, %4\n v_add_f32 %3, %4, %4" : "=v"(t[0]), "=v"(t[1]), "=v"(t[2]), "=v"(t[3]) : "v"(a[1])); ld[r] = t; }
            else if (KIND == 5) { out[(size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 4] = a[0]; *reinterpret_cast<f32x4*>(out + 4 * (size_t)((blockIdx.x * R + r) * blockDim.x + threadIdx.x)) = side[0]; ld[r] = side[r]; }
            else if (KIND == 6) { typedef float f32x2_ __attribute__((ext_vector_type(2))); const f32x2_ u0 = *reinterpret_cast<const f32x2_*>(&lds[(la + 64 * r) & 4095]); const f32x2_ u1 = *(reinterpret_cast<const f32x2_*>(&lds[(la + 64 * r + 1) & 4095]) + 1); ld[r] = f32x4{u0[0], u0[1], u1[0], u1[1]}; }
            else if (KIND == 7) { lds[(la + 64 * r) & 4095] = side[0]; ld[r] = side[r]; }
            else { f32x4 t = side[r]; asm volatile("v_fmac_f32 %0, %4, %4\n v_fmac_f32 %1, %4, %4\n v_fmac_f32 %2, %4, %4\n v_fmac_f32 %3, %4, %4" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]) : "v"(a[1])); ld[r] = t; }
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

// Role split, as in the conv kernels: waves 0-3 (one per SIMD) issue only MFMAs, waves 4-7 only the side instructions, 4 x R per MFMA-wave
// iteration of 8 MFMAs.  KIND 4: v_add_f32, 0: ds_read_b128, 5: global_store_dwordx4, 9: mixed output-pass-like (2 ds_read_b128, 2 global
// loads, 16 v_add, 2 stores).
template <int R, int KIND>
__global__ __launch_bounds__(512) void probe_split(float* out, const f32x4* w, int iters) {
    __shared__ f32x4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = f32x4{1e-3f * i, 2e-3f, 3e-3f, 4e-3f};
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        f32x16 acc[2];
        for (int m = 0; m < 2; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        float a[4], b[4];
        for (int s = 0; s < 4; ++s) { a[s] = 1e-3f * (threadIdx.x + s); b[s] = 2e-3f * (threadIdx.x - s); }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[(s + m) & 3], acc[m], 0, 0, 0);
        }
        float s = 0.f;
        for (int m = 0; m < 2; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        f32x4 side = {0.f, 0.f, 0.f, 0.f};
        float a1 = 1e-3f * threadIdx.x;
        int la = threadIdx.x;
        const f32x4* wp = w + threadIdx.x;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (KIND == 4) { f32x4 t; asm volatile("v_add_f32 %0, %4, %4\n v_add_f32 %1, %4, %4\n v_add_f32 %2, %4, %4\n v_add_f32 %3, %4, %4" : "=v"(t[0]), "=v"(t[1]), "=v"(t[2]), "=v"(t[3]) : "v"(a1)); side += t; }
                else if (KIND == 0) { side += lds[(la + 64 * r) & 4095]; }
                else if (KIND == 5) { *reinterpret_cast<f32x4*>(out + 4096 * 512 + 4 * (size_t)((blockIdx.x * R + r) * blockDim.x + threadIdx.x)) = side; }
                else { const f32x4 x0 = lds[(la + 64 * r) & 4095], x1 = lds[(la + 64 * r + 32) & 4095]; const f32x4 g0 = wp[512 * r], g1 = wp[512 * r + 256];
                       f32x4 o0 = (x0 + g0) + side, o1 = (x1 + g1) + side; o0 = o0 * 0.1f + o1; o1 = o1 * 0.1f + x0;
                       *reinterpret_cast<f32x4*>(out + 4096 * 512 + 8 * (size_t)((blockIdx.x * R + r) * blockDim.x + threadIdx.x)) = o0;
                       *reinterpret_cast<f32x4*>(out + 4096 * 512 + 8 * (size_t)((blockIdx.x * R + r) * blockDim.x + threadIdx.x) + 4) = o1; side += o1; }
            }
            la += 37;
            if ((i & 31) == 31) wp = w + threadIdx.x;
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = side[0] + side[3];
    }
}

template <int R, int KIND>
void run_split(float* out, f32x4* w) {
    const int iters = 8000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe_split<R, KIND><<<256, 512>>>(out, w, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe_split<R, KIND><<<256, 512>>>(out, w, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 8;
    printf("ROLE SPLIT: partner wave issues %s x%d per 8 MFMAs of the MFMA wave: %.2f ns per MFMA  %.1f TFLOP/s (kernel ends when the slower role ends)\n",
           KIND == 4 ? "4 x v_add_f32" : KIND == 0 ? "ds_read_b128" : KIND == 5 ? "global_store_dwordx4" : "output-pass mix", R, ms * 1e6 / mfma_per_simd,
           256.0 * 4 * mfma_per_simd * 4096 / ms / 1e9);
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
           KIND == 0 ? "ds_read_b128      " : KIND == 1 ? "global_load_dwordx4" : KIND == 2 ? "4 x v_mov_b32      " : KIND == 3 ? "2 x v_pk_add_f32   " : KIND == 4 ? "4 x v_add_f32      " : KIND == 5 ? "global_store_dwordx4" : KIND == 6 ? "2 x ds_read_b64    " : KIND == 7 ? "ds_write_b128      " : "4 x v_fmac_f32     ", R, threads / 256, ns, 256.0 * 4 * mfma_per_simd * 4096 / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, (size_t)4096 * 512 * 4 + (size_t)256 * 512 * 8 * 8 * 4 + (1 << 20));
    f32x4* w; hipMalloc(&w, 1 << 24);
    hipMemset(w, 0, 1 << 24);
    run_split<1, 4>(out, w);
    run_split<4, 4>(out, w);
    run_split<8, 4>(out, w);
    run_split<4, 0>(out, w);
    run_split<2, 5>(out, w);
    run_split<1, 9>(out, w);
    run_split<2, 9>(out, w);
    for (int threads : {256, 512}) {
        run<0, 0>(threads, out, w);
        run<2, 0>(threads, out, w);
        run<4, 0>(threads, out, w);
        run<2, 1>(threads, out, w);
        run<4, 2>(threads, out, w);
        run<4, 3>(threads, out, w);
        run<4, 4>(threads, out, w);
        run<4, 8>(threads, out, w);
        run<2, 5>(threads, out, w);
        run<4, 5>(threads, out, w);
        run<4, 6>(threads, out, w);
        run<4, 7>(threads, out, w);
    }
    return 0;
}
