// Dev tool (round 5): what does a CU's vector-memory path make of the direct-output epilogue's store patterns?
// Four waves per workgroup (one per SIMD, like the conv kernels' MFMA waves), one workgroup per CU; every wave writes its 128-row x 32-channel
// fp32 share of a tile (16 KB) NT times to fresh rows of a (rows, pitch) array, in one of these lane layouts:
//   0  x4 col-major acc  : lane (li, g) -> row li, 16 bytes at channel 8 q + 4 g          (round 4: 32 rows x 32 B per wave-store)
//   1  dword row-major   : lane (li, g) -> row 8 q + 4 g + e, channel li                   (round 5: 2 rows x 128 B per wave-store)
//   2  x4 quad-transposed: lane (c, g)  -> row 8 q + 4 g + (c & 3), 16 bytes at channel 4 (c >> 2)   (8 rows x 128 B per wave-store)
//   3  x2                : lane         -> row 8 q + 4 g + 2 (e >> 1) + (c & 1), 8 bytes at channel 2 (c >> 1)  (4 rows x 128 B)
//   4  x4 + the 4 x 4 lane transpose in registers (layout 2 produced from layout 1 with 8 DPP selects per 4 registers)
// Prints shader cycles per 16 KB wave share (median over workgroups) and bytes per clock and CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_probe/store_probe.hip -o tools/store_probe/store_probe.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* y, int pitch /*floats*/, int rows_per_wg, int nt, unsigned long long* cyc, float seed, int side_by_side) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, g = lane >> 5;
    float acc[4][16];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = seed * (float)(lane + r + 16 * mi);
    const size_t wg_row0 = (size_t)blockIdx.x * rows_per_wg;
    const unsigned pitch_b = (unsigned)pitch * 4u;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < nt; ++it) {
        // tile `it` of this workgroup: rows [it * 512 + wave * 128, +128), channel block (it % (pitch / 32))
        // side_by_side: the four waves own the four 32-channel blocks of the SAME 128 rows (the conv kernels' WN = 4 tile), else 128 rows each
        float* base = side_by_side ? y + (wg_row0 + (size_t)it * 128) * pitch + wave * 32 : y + (wg_row0 + (size_t)it * 512 + wave * 128) * pitch;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 128u * pitch_b, 0x00020000);
        if constexpr (MODE == 0) {
            const int voff = (int)(li * pitch_b + 16 * g);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_bit_cast(unsigned, acc[mi][4 * q + e] + (float)it);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff + (int)(mi * 32 * pitch_b) + 32 * q, 0, 0);
                }
        } else if constexpr (MODE == 1) {
            const int voff = (int)(4 * g * pitch_b + 4 * li);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[mi][r] + (float)it), rs,
                                                          voff + (int)((mi * 32 + 8 * (r >> 2) + (r & 3)) * pitch_b), 0, 0);
        } else if constexpr (MODE == 2 || MODE == 4) {
            const int voff = (int)((4 * g + (li & 3)) * pitch_b + 16 * (li >> 2));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float a[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = acc[mi][4 * q + e] + (float)it;
                    if constexpr (MODE == 4) {
                        // 4 x 4 transpose over the lane quad: out[lane i][j] = in[lane j][i]; two butterfly steps of DPP selects
                        const bool odd = lane & 1, hi = lane & 2;
                        float b[4];
                        {   // lanes i ^ 1: pairs (0, 1) and (2, 3)
                            const float p0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a[1]), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
                            const float p1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a[0]), 0xB1, 0xF, 0xF, true));
                            const float p2 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a[3]), 0xB1, 0xF, 0xF, true));
                            const float p3 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a[2]), 0xB1, 0xF, 0xF, true));
                            b[0] = odd ? p0 : a[0];
                            b[1] = odd ? a[1] : p1;
                            b[2] = odd ? p2 : a[2];
                            b[3] = odd ? a[3] : p3;
                        }
                        {   // lanes i ^ 2: pairs (0, 2) and (1, 3)
                            const float p0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, b[2]), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
                            const float p2 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, b[0]), 0x4E, 0xF, 0xF, true));
                            const float p1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, b[3]), 0x4E, 0xF, 0xF, true));
                            const float p3 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, b[1]), 0x4E, 0xF, 0xF, true));
                            a[0] = hi ? p0 : b[0];
                            a[2] = hi ? b[2] : p2;
                            a[1] = hi ? p1 : b[1];
                            a[3] = hi ? b[3] : p3;
                        }
                    }
                    u32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_bit_cast(unsigned, a[e]);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff + (int)((mi * 32 + 8 * q) * pitch_b), 0, 0);
                }
        } else {
            const int voff = (int)((4 * g + (li & 1)) * pitch_b + 8 * (li >> 1));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        u32x2 v;
                        v[0] = __builtin_bit_cast(unsigned, acc[mi][4 * q + 2 * h] + (float)it);
                        v[1] = __builtin_bit_cast(unsigned, acc[mi][4 * q + 2 * h + 1] + (float)it);
                        __builtin_amdgcn_raw_buffer_store_b64(v, rs, voff + (int)((mi * 32 + 8 * q + 2 * h) * pitch_b), 0, 0);
                    }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
    const int pitch = argc > 1 ? atoi(argv[1]) : 128;
    const int nt = argc > 2 ? atoi(argv[2]) : 8;
    const int nwg = argc > 3 ? atoi(argv[3]) : 256, rows_per_wg = nt * 512;
    const int sbs = argc > 4 ? atoi(argv[4]) : 0;
    float* y;
    unsigned long long* cyc;
    const size_t n = (size_t)nwg * rows_per_wg * pitch;
    hipMalloc(&y, n * 4);
    hipMalloc(&cyc, nwg * 8);
    std::vector<unsigned long long> h(nwg);
    const char* names[5] = {"x4 col-major (round 4)", "dword row-major (round 5)", "x4 quad layout (free)", "x2 pair layout (free)", "x4 + quad transpose"};
    printf("pitch %d floats, %d tiles of 512 x 32 per workgroup, %d workgroups x 4 waves%s\n", pitch, nt, nwg, sbs ? ", waves side by side in channels" : "");
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 5; ++mode) {
            for (int w = 0; w < 40; ++w) probe<1><<<nwg, 256>>>(y, pitch, rows_per_wg, nt, cyc, 0.5f, sbs);  // clock warm-up
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0);
            switch (mode) {
                case 0: probe<0><<<nwg, 256>>>(y, pitch, rows_per_wg, nt, cyc, 0.5f, sbs); break;
                case 1: probe<1><<<nwg, 256>>>(y, pitch, rows_per_wg, nt, cyc, 0.5f, sbs); break;
                case 2: probe<2><<<nwg, 256>>>(y, pitch, rows_per_wg, nt, cyc, 0.5f, sbs); break;
                case 3: probe<3><<<nwg, 256>>>(y, pitch, rows_per_wg, nt, cyc, 0.5f, sbs); break;
                default: probe<4><<<nwg, 256>>>(y, pitch, rows_per_wg, nt, cyc, 0.5f, sbs); break;
            }
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), cyc, nwg * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            const double med = (double)h[nwg / 2] / nt;  // cycles (100 MHz-independent: s_memtime) per 64 KB workgroup tile
            printf("  %-28s %8.0f cycles per 64-KB tile (median WG)  = %5.1f B/clk/CU   launch %.1f us = %.2f TB/s\n", names[mode], med, 65536.0 / med, ms * 1e3,
                   (double)n * 4 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
