// Dev tool (round 5): what would ONE persistent launch per ResBlock stage buy at small batches?
// The batch-1 AR step is 24 dependent conv launches of 10-25 us.  A launch covering a stage's six layers replaces five kernel
// boundaries by in-launch hand-offs: a workgroup owns a (branch, row tile) through all layers, publishes its output tile (write-through sc1 stores +
// vmcnt(0) + a flag) and the neighbouring row tiles' owners wait for the flag before they stage the rows they need (halo) by LDS-DMA.  No grid barrier:
// a tile only waits for its two neighbours.  This probe runs exactly that data flow with the real kernel's geometry — 8-wave workgroups (4 MFMA + 4 loader
// waves), 128-row x 32-channel fp32 tiles (16 KB), 228 staged rows per tile (50-row halos), a dependent chain of fp32 MFMAs standing in for the K loop —
// once as L dependent launches and once as L phases of one launch, and checks every staged halo value (a stale read shows up as an error count).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/chain_probe.hip -o tools/chain_probe.bin && timeout 120 tools/chain_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kRows = 128, kC = 32, kHalo = 50;           // tile geometry (stage 3 of HiFi-CAR: C = 32)
constexpr int kTileFloats = kRows * kC;                    // 16 KB
constexpr int kStageRows = kRows + 2 * kHalo;              // 228 rows staged per tile
constexpr int kThreads = 512;

struct Args {
    float* buf[2];         // ping-pong activations: [tile][row][channel]
    unsigned* flags;       // [tile]: layers published so far (monotonic over the whole run)
    unsigned* errors;
    int tiles_per_branch;  // neighbours exist inside a branch only
    int mfma_per_wave;     // length of the stand-in K loop
    unsigned base;         // flag value before this launch's first layer
};

__device__ __forceinline__ void dma16(const char* src, unsigned lds_addr, bool sc1) {
    if (sc1) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1" : : "s"(lds_addr), "v"(src) : "memory");
    else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" : : "s"(lds_addr), "v"(src) : "memory");
}

// one layer of one tile.  CHAIN: wait for the neighbours' flags first, write-through stores, publish a flag.
template <bool CHAIN>
__device__ __forceinline__ void layer_body(const Args& a, int layer, int tile, char* smem, unsigned long long* stamps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* src = a.buf[layer & 1];
    float* dst = a.buf[(layer + 1) & 1];
    const int tb = tile % a.tiles_per_branch;
    const bool has_l = tb > 0, has_r = tb + 1 < a.tiles_per_branch;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    if (wave >= 4) {  // ---- loader waves: (wait,) stage rows [-50, 178) of the tile
        if (CHAIN && layer > 0) {  // every loader wave's lane 0 polls (no extra intra-workgroup signal needed before the DMA issue)
            if (lane == 0) {
                const unsigned want = a.base + (unsigned)layer;
                if (has_l) while (__hip_atomic_load(a.flags + tile - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
                if (has_r) while (__hip_atomic_load(a.flags + tile + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
            }
        }
        if (stamps && wave == 4 && lane == 0) stamps[1] = __builtin_amdgcn_s_memrealtime();
        const int lw = wave - 4;
        constexpr int kInstr = kStageRows * kC * 4 / 1024;  // 1-KB pieces: 28.5 -> 29
        for (int i = lw; i < (kStageRows * kC * 4 + 1023) / 1024; i += 4) {
            // piece i covers staged bytes [1024 i, 1024 i + 1024): row = byte / 128 - 50
            const int byte = i * 1024 + lane * 16;
            const int row = byte / (kC * 4) - kHalo;
            const int col = byte % (kC * 4);
            const bool own = row >= 0 && row < kRows;
            const char* p = reinterpret_cast<const char*>(src) + ((long long)tile * kRows + row) * (kC * 4) + col;
            const bool exists = byte < kStageRows * kC * 4 && (own || (row < 0 ? has_l : has_r));
            if (!exists) p = reinterpret_cast<const char*>(a.buf[0]);  // (any readable address: the value is not checked)
            dma16(p, lds0 + (unsigned)i * 1024u, CHAIN);
        }
        (void)kInstr;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();  // staged
    if (stamps && tid == 0) stamps[2] = __builtin_amdgcn_s_memrealtime();
    // ---- check the halo rows: every value of layer l's input must be (float)(base_layer + l) written by the previous layer
    if (layer > 0) {
        const float want = (float)layer;
        unsigned bad = 0;
        for (int i = tid; i < kStageRows * kC; i += kThreads) {
            const int row = i / kC - kHalo;
            const bool exists = (row >= 0 && row < kRows) || (row < 0 ? has_l : has_r);
            if (exists && reinterpret_cast<const float*>(smem)[i] != want) ++bad;
        }
        if (bad) atomicAdd(a.errors, bad);
    }
    if (wave < 4) {  // ---- MFMA waves: a dependent-accumulator-free chain as the K loop, then the tile's epilogue
        f32x16 acc[4];
        for (int m = 0; m < 4; ++m)
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        const float x = reinterpret_cast<const float*>(smem)[tid], w = 1e-3f * lane;
        for (int i = 0; i < a.mfma_per_wave; i += 4)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, x, acc[m], 0, 0, 0);
        float keep = 0.f;
        for (int m = 0; m < 4; ++m) keep += acc[m][0];
        // epilogue: wave w stores rows [32 w, 32 w + 32): 4 KB = 4 x 16-byte stores per lane
        const float val = (float)(layer + 1) + (keep == 12345.678f ? 1.f : 0.f);
        float* row0 = dst + ((long long)tile * kRows + wave * 32) * kC;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)row0, 0, 32 * kC * 4, 0x00020000);
        const f32x4 v = {val, val, val, val};
#pragma unroll
        for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (q * 64 + lane) * 16, 0, CHAIN ? 16 : 0);  // aux 16 = sc1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (stamps && tid == 0) stamps[3] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();  // every wave's stores have left; the staging buffer is free
    if (CHAIN && tid == 0) __hip_atomic_store(a.flags + tile, a.base + (unsigned)layer + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (stamps && tid == 0) stamps[4] = __builtin_amdgcn_s_memrealtime();
}

__global__ __launch_bounds__(kThreads) void layer_kernel(const Args a, int layer) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    layer_body<false>(a, layer, blockIdx.x, smem, nullptr);
}
__global__ __launch_bounds__(kThreads) void chain_kernel(const Args a, int layers, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    for (int l = 0; l < layers; ++l) {
        unsigned long long* st = (stamps && blockIdx.x == 1) ? stamps + l * 8 : nullptr;
        if (st && threadIdx.x == 0) st[0] = __builtin_amdgcn_s_memrealtime();
        layer_body<true>(a, l, blockIdx.x, smem, st);
    }
}

int main(int argc, char** argv) {
    const int layers = 6, reps = 200;
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds = 32 * 1024;
    printf("six dependent layers of 128-row x 32-channel tiles (16 KB out, 29 KB staged per tile), %d repetitions\n", reps);
    for (int mf : {48, 112, 176}) {  // 3 / 7 / 11 taps x 2 slabs x 8 MFMAs per 32-row block
        for (int nwg : {16, 48, 96, 256}) {
            Args a;
            const size_t n = (size_t)nwg * kTileFloats;
            hipMalloc(&a.buf[0], n * 4);
            hipMalloc(&a.buf[1], n * 4);
            hipMalloc(&a.flags, nwg * 4);
            hipMalloc(&a.errors, 4);
            hipMemset(a.buf[0], 0, n * 4);
            hipMemset(a.buf[1], 0, n * 4);
            hipMemset(a.flags, 0, nwg * 4);
            hipMemset(a.errors, 0, 4);
            a.tiles_per_branch = 16;
            a.mfma_per_wave = mf;
            a.base = 0;
            unsigned long long* stamps;
            hipMalloc(&stamps, layers * 8 * 8);
            hipMemset(stamps, 0, layers * 8 * 8);
            float ms_l = 0, ms_c = 0;
            for (int mode = 0; mode < 2; ++mode) {
                for (int pass = 0; pass < 2; ++pass) {  // pass 0 warms up (clock ramp)
                    hipEventRecord(e0, s);
                    for (int r = 0; r < reps; ++r) {
                        if (mode == 0) {
                            for (int l = 0; l < layers; ++l) hipLaunchKernelGGL(layer_kernel, dim3(nwg), dim3(kThreads), lds, s, a, l);
                        } else {
                            hipLaunchKernelGGL(chain_kernel, dim3(nwg), dim3(kThreads), lds, s, a, layers, (pass == 1 && r == reps - 1) ? stamps : nullptr);
                            a.base += layers;
                        }
                    }
                    hipEventRecord(e1, s);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(mode ? &ms_c : &ms_l, e0, e1);
                }
            }
            unsigned err = 0;
            hipMemcpy(&err, a.errors, 4, hipMemcpyDeviceToHost);
            unsigned long long hs[6 * 8];
            hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost);
            printf("K loop %3d MFMAs/wave, %3d workgroups: launches %6.2f us per layer | one launch per stage %6.2f us per layer (%+5.1f %%)  stale/wrong values: %u\n", mf, nwg,
                   ms_l * 1e3 / (reps * layers), ms_c * 1e3 / (reps * layers), 100.0 * (ms_l / ms_c - 1.0), err);
            if (nwg == 48) {
                printf("    workgroup 1 in the chained launch, per layer [wait for flags | staging | K loop + stores | publish] us: ");
                for (int l = 1; l < layers; ++l)
                    printf("%.2f|%.2f|%.2f|%.2f  ", (hs[l * 8 + 1] - hs[l * 8 + 0]) * 0.01, (hs[l * 8 + 2] - hs[l * 8 + 1]) * 0.01, (hs[l * 8 + 3] - hs[l * 8 + 2]) * 0.01,
                           (hs[l * 8 + 4] - hs[l * 8 + 3]) * 0.01);
                printf("\n");
            }
            hipFree(a.buf[0]); hipFree(a.buf[1]); hipFree(a.flags); hipFree(a.errors); hipFree(stamps);
        }
    }
    return 0;
}
