// Backward (training) kernels of the HiFi-GAN / HiFi-CAR generator on gfx950: SURVEY.md section 8 row f1, generator part.
//
// The reference trains with PyTorch autograd through torch.nn.Conv1d / ConvTranspose1d / Linear (articulatory/bin/train.py:241-440,
// articulatory/models/hifigan.py:198-239).  Per conv layer y = conv(a, W) + b with a = LeakyReLU(x):
//   data gradient    da = conv(dy, W flipped and transposed)   -> the forward conv kernels on a second weight pack, with the
//                                                                 LeakyReLU'(a) mask and the skip gradient fused into the output pass
//                                                                 (ConvParams::mask_src); a ConvTranspose1d's data gradient is a plain
//                                                                 3-tap Conv1d over its phase-major "virtual channel" rows
//   weight gradient  dW[co, ci, k] = sum_{b,t} dy[b, t, co] * a[b, t + off_k, ci]   -> wgrad_kernel (MFMA, reduction over rows)
//   bias gradient    db[co] = sum_{b,t} dy[b, t, co]                                -> column sums of the weight-gradient kernels' staged tiles
// plus the small ends of the network (output conv + tanh, MRF mean, feature transpose, PastFCEncoder) as VALU kernels.
// All reductions are two-stage (partials per row split, then a fixed-order sum): deterministic, no atomics.
#pragma once
#include "hificar_kernels.hip.h"

namespace hificar {

// ------------------------------------------------------------------------------------------------
// Weight gradient: P[split][tap][g][a] = sum over the split's rows of G[row, g] * A[row + off(tap, phase(g)), a]
//   G: upstream gradient rows (nseq, L, gpitch);  A: activated layer input rows (nseq, L, apitch); rows outside [0, L) are zero.
// One workgroup = 4 waves in a 2 x 2 grid, each wave one 32 x 32 block (v_mfma_f32_32x32x2_f32: the row axis is the MFMA's K),
// i.e. a 64 (g) x 64 (a) tile of one tap; the rows come in chunks of 128 through LDS, the next chunk's global loads in flight in
// registers while the current one is multiplied.
// ------------------------------------------------------------------------------------------------
struct WgradParams {
    const float* g;
    const float* a;
    float* partial;  // [nsplit][ntaps][gpad][apad], gpad = n_gblk * 32, apad = n_ablk * 32
    int nseq, L;
    int gpitch, apitch;
    int n_gblk, n_ablk;
    int ntaps;
    int tap_step;
    int tap_off0[kMaxPhase];
    int nb32_per_phase;  // g blocks per phase (ConvTranspose1d: output phase r owns blocks [r * nb32_per_phase, ...))
    int chunks_per_seq;  // ceil(L / 128)
    int nsplit;
    int g_per_tile;      // 32-blocks of g per workgroup tile: 2, or 1 when a 64-wide tile would straddle two phases (odd nb32_per_phase)
    int row_split;       // 1, 2 or 4: layers narrower than the 64 x 64 workgroup tile give the spare waves a share of each chunk's rows
                         // (summed inside the workgroup before the partial is written)
    float* bias_partial; // [nsplit][gpad] column sums of G (bias gradient partials), taken from the staged G tile by the workgroups of
                         // tap 0 / first a-tile; null: not wanted
};

constexpr int kWgR = 128;  // rows per chunk

__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
    __shared__ __attribute__((aligned(16))) float gs[kWgR][64];
    __shared__ __attribute__((aligned(16))) float as[kWgR][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hf = lane >> 5;
    const int gpt = p.g_per_tile;
    const int gt_n = (p.n_gblk + gpt - 1) / gpt, at_n = (p.n_ablk + 1) >> 1;
    int tile = blockIdx.x;
    const int at = tile % at_n;
    tile /= at_n;
    const int gt = tile % gt_n;
    const int tap = tile / gt_n;
    const int split = blockIdx.y;
    // wave -> (block of the tile, share of the rows): 4 blocks x 1 share, 2 x 2 or 1 x 4
    const int rs = p.row_split, nblk = 4 / rs;
    const int bi = wave % nblk, part = wave / nblk;
    const int ab_w = p.n_ablk >= 2 ? 2 : 1;  // blocks along a inside the tile
    const int gl = bi / ab_w, al = bi % ab_w;
    const int gblk = gt * gpt + gl, ablk = at * 2 + al;
    const bool active = gl < gpt && gblk < p.n_gblk && ablk < p.n_ablk;
    // the g blocks of a tile share one tap offset (the staged A rows are shifted by it): g_per_tile = 1 when phases are 32 wide
    const int phase = (gt * gpt) / p.nb32_per_phase;
    const int off = p.tap_off0[phase] + tap * p.tap_step;
    const int nchunks = p.nseq * p.chunks_per_seq;
    const int c_lo = (int)((long long)nchunks * split / p.nsplit), c_hi = (int)((long long)nchunks * (split + 1) / p.nsplit);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // staging: thread -> 16 float4 per operand tile of 128 rows x 64 channels (2048 float4): index q * 256 + tid
    f32x4 gr[8], ar[8];
    auto fetch = [&](int c) {
        const int seq = c / p.chunks_per_seq;
        const int t0 = (c - seq * p.chunks_per_seq) * kWgR;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int idx = q * 256 + tid;  // float4 index: row = idx / 16, channel group = idx % 16
            const int row = idx >> 4, c4 = (idx & 15) * 4;
            const int tg = t0 + row, ta = tg + off;
            f32x4 vg = {0.f, 0.f, 0.f, 0.f}, va = {0.f, 0.f, 0.f, 0.f};
            const int gch = gt * gpt * 32 + c4, ach = at * 64 + c4;
            if (tg < p.L && c4 < gpt * 32 && gch < p.gpitch) vg = *reinterpret_cast<const f32x4*>(p.g + ((size_t)seq * p.L + tg) * p.gpitch + gch);
            if (tg < p.L && ta >= 0 && ta < p.L && ach < p.apitch) va = *reinterpret_cast<const f32x4*>(p.a + ((size_t)seq * p.L + ta) * p.apitch + ach);
            gr[q] = vg;
            ar[q] = va;
        }
    };
    const bool do_bias = p.bias_partial && at == 0 && tap == 0;
    float bsum = 0.f;  // thread (channel tid & 63, row slice tid >> 6) of the G tile
    if (c_lo < c_hi) fetch(c_lo);
    for (int c = c_lo; c < c_hi; ++c) {
        __syncthreads();  // the previous chunk's reads are done
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int idx = q * 256 + tid;
            const int row = idx >> 4, c4 = (idx & 15) * 4;
            *reinterpret_cast<f32x4*>(&gs[row][c4]) = gr[q];
            *reinterpret_cast<f32x4*>(&as[row][c4]) = ar[q];
        }
        __syncthreads();
        if (c + 1 < c_hi) fetch(c + 1);  // in flight while this chunk is multiplied
        if (do_bias) {
#pragma unroll 8
            for (int r = tid >> 6; r < kWgR; r += 4) bsum += gs[r][tid & 63];
        }
        if (active) {
            const int gc = gl * 32 + li, ac = al * 32 + li;
            const int k0 = part * (kWgR / rs);
            if (rs == 1) {
#pragma unroll 8
                for (int k = 0; k < kWgR; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gs[k + hf][gc], as[k + hf][ac], acc, 0, 0, 0);
            } else if (rs == 2) {
#pragma unroll 8
                for (int k = 0; k < kWgR / 2; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gs[k0 + k + hf][gc], as[k0 + k + hf][ac], acc, 0, 0, 0);
            } else {
#pragma unroll 8
                for (int k = 0; k < kWgR / 4; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gs[k0 + k + hf][gc], as[k0 + k + hf][ac], acc, 0, 0, 0);
            }
        }
    }
    if (do_bias) {
        __syncthreads();
        as[tid >> 6][tid & 63] = bsum;  // reuse the A tile as scratch
        __syncthreads();
        const int ch = gt * gpt * 32 + tid;
        if (tid < gpt * 32 && ch < p.n_gblk * 32) p.bias_partial[(size_t)split * p.n_gblk * 32 + ch] = (as[0][tid] + as[1][tid]) + (as[2][tid] + as[3][tid]);
    }
    if (rs > 1) {  // the row shares of a block are summed in the workgroup, in the fixed order part 0 + 1 (+ 2 + 3)
        __syncthreads();
        float* red = &gs[0][0];  // 8192 floats; needed: (rs - 1) * nblk * 1024 <= 3072
        if (active && part > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((part - 1) * nblk + bi) * 1024 + r * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (active && part == 0) {
            for (int q = 1; q < rs; ++q) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += red[((q - 1) * nblk + bi) * 1024 + r * 64 + lane];
            }
        }
    }
    if (active && part == 0) {
        const int gpad = p.n_gblk * 32, apad = p.n_ablk * 32;
        float* dst = p.partial + (((size_t)split * p.ntaps + tap) * gpad + gblk * 32) * apad + ablk * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(size_t)((r & 3) + 8 * (r >> 2) + 4 * hf) * apad] = acc[r];
    }
}

// All taps from one staging: the per-tap kernels above re-read both operand tiles once per tap (16 GB of L2 / MALL traffic per backward
// of the batch-64 step: they are bound by it, 51-54 TFLOP/s).  Here a workgroup stages a chunk of G rows and the matching A rows WITH the
// taps' halo once (LDS-DMA, double-buffered, no staging registers) and keeps one 32 x 32 accumulator PER TAP in every wave: a K step reads
// one G operand and one A operand per tap (row-shifted) from LDS and issues NT MFMAs.  Workgroup tile 64 (g) x 64 (a), waves 2 x 2;
// partial layout, row shares and bias sums as above.  NT = accumulators instantiated (>= ntaps).
struct WgradTapsParams {
    long long zs_g = 0, zs_a = 0, zs_partial = 0, zs_bias = 0;  // per-group strides (floats)
    // A-operand addressing: row t of sequence s at a + s * a_seq_pitch + t * apitch, acols valid columns.  0 = packed rows
    // (a_seq_pitch = L * apitch, acols = apitch); a pitch shorter than the row = the sliding-window view of ConvParams::x_row_bytes
    long long a_seq_pitch = 0;
    int acols = 0;
    int a_rows = 0;  // A rows that exist per sequence (0: L, as G): the polyphase-input form of a strided conv reads ntaps - 1 rows past L
    WgradParams w;
    const char* zeros;  // >= 16 zero bytes (rows outside the sequence)
    int off_min;        // smallest tap offset of the layer (over all phases), halo = largest - smallest
    int halo;
};

constexpr int kWgtR = 64;  // G rows per chunk (default; wgrad_chunk_rows picks 32 / 64 / 128 per layer)

// EXACT: the layer has exactly NT taps (no per-tap branch in the K loop: the operand reads of the next row pair are in flight while this
// pair's MFMAs issue); otherwise ntaps < NT and the surplus taps are skipped.
// Up to two layers of the same tap count per launch (blockIdx.z): conv2 and conv1 of a ResBlock slot have independent weight gradients,
// and one launch with half the row splits each fills the chip with half the partial sums to write and reduce.
// A grouped conv's groups (discriminators) ride on blockIdx.z as well: entry e = blockIdx.z / zg, group z = blockIdx.z % zg, whose
// operands / partials lie z strides further on.
struct WgradTapsPair {
    WgradTapsParams q[2];
    int zg;
};

// R: G rows per chunk (wgrad_chunk_rows: 32 for sequences of at most 32 rows, 128 where few taps leave a 64-row chunk too little work per
// barrier, else 64)
template <int NT, bool EXACT, int R>
__global__ __launch_bounds__(256) void wgrad_taps_kernel(const WgradTapsPair pair) {
    extern __shared__ __attribute__((aligned(1024))) char wgt_smem[];
    const WgradTapsParams& q = pair.q[blockIdx.z / pair.zg];
    const int zi = blockIdx.z % pair.zg;
    const float* const g_base = q.w.g + (size_t)zi * q.zs_g;
    const float* const a_base = q.w.a + (size_t)zi * q.zs_a;
    float* const partial_base = q.w.partial + (size_t)zi * q.zs_partial;
    float* const bias_base = q.w.bias_partial ? q.w.bias_partial + (size_t)zi * q.zs_bias : nullptr;
    const WgradParams& p = q.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hf = lane >> 5;
    const int gpt = p.g_per_tile;
    const int at_n = (p.n_ablk + 1) >> 1;
    const int at = blockIdx.x % at_n, gt = blockIdx.x / at_n;
    const int split = blockIdx.y;
    const int rs = p.row_split, nblk = 4 / rs;
    const int bi = wave % nblk, part = wave / nblk;
    const int ab_w = p.n_ablk >= 2 ? 2 : 1;
    const int gl = bi / ab_w, al = bi % ab_w;
    const int gblk = gt * gpt + gl, ablk = at * 2 + al;
    const bool active = gl < gpt && gblk < p.n_gblk && ablk < p.n_ablk;
    const int phase = (gt * gpt) / p.nb32_per_phase;
    const int off0 = p.tap_off0[phase] - q.off_min;  // row shift of tap 0 inside the staged A tile (>= 0)
    const int a_rows = R + q.halo;
    const int g_bytes = R * 256, a_bytes = ((a_rows * 256 + 1023) >> 10) << 10, buf_bytes = g_bytes + a_bytes;
    const int cps = (p.L + R - 1) / R;
    const int nchunks = p.nseq * cps;
    const int a_cols = q.acols ? q.acols : p.apitch;
    const size_t a_seq_pitch = q.a_seq_pitch ? (size_t)q.a_seq_pitch : (size_t)p.L * p.apitch;
    const int c_lo = (int)((long long)nchunks * split / p.nsplit), c_hi = (int)((long long)nchunks * (split + 1) / p.nsplit);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // LDS-DMA of one chunk: 1 KiB = 4 rows of 64 channels per wave-instruction, lane -> (row 4 i + lane / 16, channels 4 (lane % 16) ..)
    auto stage = [&](int c, int b) {
        const int seq = c / cps;
        const int t0 = (c - seq * cps) * R;
        char* dst = wgt_smem + b * buf_bytes;
        const int c4 = (lane & 15) * 4;
        for (int i = wave; i < R / 4; i += 4) {
            const int tg = t0 + 4 * i + (lane >> 4);
            const char* src = q.zeros;
            const int gch = gt * gpt * 32 + c4;
            if (tg < p.L && c4 < gpt * 32 && gch < p.gpitch) src = reinterpret_cast<const char*>(g_base + ((size_t)seq * p.L + tg) * p.gpitch + gch);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        }
        const int ninstr = (a_rows + 3) >> 2;
        for (int i = wave; i < ninstr; i += 4) {
            const int r = 4 * i + (lane >> 4);
            const int ta = t0 + q.off_min + r;
            const char* src = q.zeros;
            const int ach = at * 64 + c4;
            // an A row is only ever multiplied with G rows of the same chunk: rows whose G partner lies beyond the sequence need no masking
            // (those G rows are zero), but A rows outside [0, L) are the conv's zero padding
            if (r < a_rows && ta >= 0 && ta < (q.a_rows ? q.a_rows : p.L) && ach < a_cols)
                src = reinterpret_cast<const char*>(a_base + (size_t)seq * a_seq_pitch + (size_t)ta * p.apitch + ach);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + g_bytes + i * 1024), 16, 0, 0);
        }
    };
    const bool do_bias = bias_base && at == 0;
    float bsum = 0.f;
    if (c_lo < c_hi) stage(c_lo, 0);
    __syncthreads();  // (hipcc drains vmcnt before the barrier)
    for (int c = c_lo; c < c_hi; ++c) {
        const int b = (c - c_lo) & 1;
        if (c + 1 < c_hi) stage(c + 1, b ^ 1);  // lands while this chunk is multiplied
        const float* gs = reinterpret_cast<const float*>(wgt_smem + b * buf_bytes);
        const float* as = gs + g_bytes / 4;
        if (do_bias) {
#pragma unroll 8
            for (int r = tid >> 6; r < R; r += 4) bsum += gs[r * 64 + (tid & 63)];
        }
        if (active) {
            const int gc = gl * 32 + li, ac = al * 32 + li;
            const int k0 = part * (R / rs), kn = R / rs;  // kn: 64, 32 or 16 rows
            const int tstep = p.tap_step * 64;
            const float* gp = gs + (k0 + hf) * 64 + gc;
            const float* ap[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) ap[t] = as + (k0 + hf + off0) * 64 + ac + ((EXACT || t < p.ntaps) ? t * tstep : 0);
            // 16 rows per trip, unrolled (row offsets are immediates of the reads); operands of the next row pair are fetched into the
            // other register set while this pair's MFMAs issue.  The last fetch of a chunk reads two rows past it (inside the LDS
            // allocation, never used).
            float g0 = gp[0], g1, a0[NT], a1[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) a0[t] = ap[t][0];
            for (int kb = 0; kb < kn; kb += 16) {
#pragma unroll
                for (int k = 0; k < 16; k += 4) {
                    g1 = gp[(k + 2) * 64];
#pragma unroll
                    for (int t = 0; t < NT; ++t) a1[t] = ap[t][(k + 2) * 64];
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (EXACT || t < p.ntaps) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, a0[t], acc[t], 0, 0, 0);
                    g0 = gp[(k + 4) * 64];
#pragma unroll
                    for (int t = 0; t < NT; ++t) a0[t] = ap[t][(k + 4) * 64];
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (EXACT || t < p.ntaps) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, a1[t], acc[t], 0, 0, 0);
                }
                gp += 16 * 64;
#pragma unroll
                for (int t = 0; t < NT; ++t) ap[t] += 16 * 64;
            }
        }
        __syncthreads();  // next chunk landed, everyone done with this one
    }
    if (do_bias) {
        float* red = reinterpret_cast<float*>(wgt_smem);
        red[(tid >> 6) * 64 + (tid & 63)] = bsum;
        __syncthreads();
        const int ch = gt * gpt * 32 + tid;
        if (tid < gpt * 32 && ch < p.n_gblk * 32) bias_base[(size_t)split * p.n_gblk * 32 + ch] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
        __syncthreads();
    }
    const int gpad = p.n_gblk * 32, apad = p.n_ablk * 32;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t >= p.ntaps) break;
        f32x16 v = acc[t];
        if (rs > 1) {  // row shares of this tap summed in the workgroup, fixed order
            float* red = reinterpret_cast<float*>(wgt_smem);
            if (active && part > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((part - 1) * nblk + bi) * 1024 + r * 64 + lane] = v[r];
            }
            __syncthreads();
            if (active && part == 0) {
                for (int qq = 1; qq < rs; ++qq)
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] += red[((qq - 1) * nblk + bi) * 1024 + r * 64 + lane];
            }
            __syncthreads();
        }
        if (active && part == 0) {
            float* dst = partial_base + (((size_t)split * p.ntaps + t) * gpad + gblk * 32) * apad + ablk * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(size_t)((r & 3) + 8 * (r >> 2) + 4 * hf) * apad] = v[r];
        }
    }
}

// One-tap form with a 128 x 128 tile (the discriminators' GEMM-form convs: dW [N x Kg] = dZ^T A with few rows M and wide Kg): a wave owns
// a 64 x 64 block as 2 x 2 accumulators, so a 64-row chunk staged once (64 KB) feeds 128 MFMAs per wave instead of the 32 of the
// 64 x 64 tile above, which is bound by the staging traffic at one tap.  Same partial layout, bias sums and z replicas.
// AJ = 4 (round 4): a 128 x 256 tile — a wave owns 64 x 128 as 2 x 4 accumulators (128 registers; one workgroup per CU, so a wave has 512) and
// the chunk is 32 rows: 48 KB staged per 128 MFMAs per wave instead of 64 KB (the 128 x 128 tile pulls 8 B per clock and CU through the L2 /
// MALL: 0.51 of the MFMA peak measured, next to 0.64 for the forward GEMMs of the same layers), six 4-byte LDS reads per eight MFMAs instead of
// four per four.  For layers at least 256 columns wide (wgrad_gemm_wide).
constexpr int kWggR = 64;

template <int AJ, int R>
__global__ __launch_bounds__(256) void wgrad_gemm_kernel(const WgradTapsPair pair) {
    static_assert((AJ == 2 && R == 64) || (AJ == 4 && R == 32), "128 x 128 tile on 64-row chunks, or 128 x 256 on 32-row chunks");
    constexpr int AW = AJ * 64;  // columns of the staged A tile
    extern __shared__ __attribute__((aligned(1024))) char wgg_smem[];
    const WgradTapsParams& q = pair.q[blockIdx.z / pair.zg];
    const int zi = blockIdx.z % pair.zg;
    const float* const g_base = q.w.g + (size_t)zi * q.zs_g;
    const float* const a_base = q.w.a + (size_t)zi * q.zs_a;
    float* const partial_base = q.w.partial + (size_t)zi * q.zs_partial;
    float* const bias_base = q.w.bias_partial ? q.w.bias_partial + (size_t)zi * q.zs_bias : nullptr;
    const WgradParams& p = q.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hf = lane >> 5;
    const int at_n = (p.n_ablk + 2 * AJ - 1) / (2 * AJ);
    const int at = blockIdx.x % at_n, gt = blockIdx.x / at_n;
    const int split = blockIdx.y;
    const int gw = wave >> 1, aw = wave & 1;
    constexpr int buf_floats = R * (128 + AW);  // G rows then A rows
    const int cps = (p.L + R - 1) / R;
    const int nchunks = p.nseq * cps;
    const int a_cols = q.acols ? q.acols : p.apitch;
    const size_t a_seq_pitch = q.a_seq_pitch ? (size_t)q.a_seq_pitch : (size_t)p.L * p.apitch;
    const int c_lo = (int)((long long)nchunks * split / p.nsplit), c_hi = (int)((long long)nchunks * (split + 1) / p.nsplit);
    f32x16 acc[2][AJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < AJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // LDS-DMA: 1 KiB per wave-instruction = 2 rows of 128 channels (lane -> row 2 i + lane / 32, channels 4 (lane % 32) ..) or 1 row of 256
    auto stage = [&](int c, int b) {
        const int seq = c / cps;
        const int t0 = (c - seq * cps) * R;
        char* dst = wgg_smem + (size_t)b * buf_floats * 4;
        const int c4 = (lane & 31) * 4;
        for (int i = wave; i < R / 2; i += 4) {
            const int t = t0 + 2 * i + (lane >> 5);
            const char* gsrc = q.zeros;
            const int gch = gt * 128 + c4;
            if (t < p.L && gch < p.gpitch) gsrc = reinterpret_cast<const char*>(g_base + ((size_t)seq * p.L + t) * p.gpitch + gch);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        }
        if constexpr (AJ == 2) {
            for (int i = wave; i < R / 2; i += 4) {
                const int t = t0 + 2 * i + (lane >> 5);
                const char* asrc = q.zeros;
                const int ach = at * AW + c4;
                if (t < p.L && ach < a_cols) asrc = reinterpret_cast<const char*>(a_base + (size_t)seq * a_seq_pitch + (size_t)t * p.apitch + ach);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)asrc,
                                                 (__attribute__((address_space(3))) void*)(dst + R * 512 + i * 1024), 16, 0, 0);
            }
        } else {
            for (int i = wave; i < R; i += 4) {
                const int t = t0 + i;
                const char* asrc = q.zeros;
                const int ach = at * AW + lane * 4;
                if (t < p.L && ach < a_cols) asrc = reinterpret_cast<const char*>(a_base + (size_t)seq * a_seq_pitch + (size_t)t * p.apitch + ach);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)asrc,
                                                 (__attribute__((address_space(3))) void*)(dst + R * 512 + i * 1024), 16, 0, 0);
            }
        }
    };
    const bool do_bias = bias_base && at == 0;
    float bsum = 0.f;
    if (c_lo < c_hi) stage(c_lo, 0);
    __syncthreads();
    for (int c = c_lo; c < c_hi; ++c) {
        const int b = (c - c_lo) & 1;
        if (c + 1 < c_hi) stage(c + 1, b ^ 1);
        const float* gs = reinterpret_cast<const float*>(wgg_smem) + (size_t)b * buf_floats;
        const float* as = gs + R * 128;
        if (do_bias) {
#pragma unroll 8
            for (int r = tid >> 7; r < R; r += 2) bsum += gs[r * 128 + (tid & 127)];
        }
        {
            const float* gp = gs + hf * 128 + gw * 64 + li;
            const float* ap = as + hf * AW + aw * (AJ * 32) + li;
            float g0[2] = {gp[0], gp[32]}, a0[AJ], g1[2], a1[AJ];
#pragma unroll
            for (int j = 0; j < AJ; ++j) a0[j] = ap[32 * j];
#pragma unroll
            for (int kb = 0; kb < R; kb += 16) {
#pragma unroll
                for (int k = 0; k < 16; k += 4) {
                    const int r1 = kb + k + 2, r2 = kb + k + 4;  // (the last fetch reads two rows past the chunk: unused)
                    g1[0] = gp[r1 * 128], g1[1] = gp[r1 * 128 + 32];
#pragma unroll
                    for (int j = 0; j < AJ; ++j) a1[j] = ap[r1 * AW + 32 * j];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < AJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0[i], a0[j], acc[i][j], 0, 0, 0);
                    g0[0] = gp[r2 * 128], g0[1] = gp[r2 * 128 + 32];
#pragma unroll
                    for (int j = 0; j < AJ; ++j) a0[j] = ap[r2 * AW + 32 * j];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < AJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1[i], a1[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    if (do_bias) {
        float* red = reinterpret_cast<float*>(wgg_smem);
        red[(tid >> 7) * 128 + (tid & 127)] = bsum;
        __syncthreads();
        const int ch = gt * 128 + tid;
        if (tid < 128 && ch < p.n_gblk * 32) bias_base[(size_t)split * p.n_gblk * 32 + ch] = red[tid] + red[128 + tid];
    }
    const int gpad = p.n_gblk * 32, apad = p.n_ablk * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int gblk = gt * 4 + gw * 2 + i, ablk = at * (2 * AJ) + aw * AJ + j;
            if (gblk >= p.n_gblk || ablk >= p.n_ablk) continue;
            float* dst = partial_base + ((size_t)split * gpad + gblk * 32) * apad + ablk * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(size_t)((r & 3) + 8 * (r >> 2) + 4 * hf) * apad] = acc[i][j][r];
        }
}

// dW in the reference's layout from the partials, summed over the splits in a fixed order.
//   Conv1d          : dst[(co * cin + ci) * K + k]     g = co, a = ci, k = tap
//   ConvTranspose1d : dst[(ci * cout + co) * K + k]    g = r * cout_pad + co (phase-major), a = ci, k = tap_k[r][tap] (< 0: no such weight)
struct WreduceParams {
    const float* partial;
    float* dst;
    int nsplit, ntaps, gpad, apad;
    int cout, cin, K;
    int transposed, n_phase, cout_pad;
    int tap_k[kMaxPhase][kMaxTaps];
    int flip;  // 0
    int gemm_cin;  // > 0: GEMM form of a conv (discriminators): a = tap * gemm_cin + c -> dst[(co * gemm_cin + c) * K + tap]
    int accumulate;  // 1: dst += the reduced gradient (a second backward pass into the same buffer) instead of dst = it
    int poly_s, poly_cin;  // poly_s > 0: polyphase-input form of a strided conv (cout, poly_cin, K): (tap q, a = r * poly_cin + c) -> kernel index
                           // poly_s * q + r, dst[(co * poly_cin + c) * K + poly_s * q + r] (indices >= K carry zero weights: skipped)
};

struct WreducePair {
    WreduceParams r[2];
    int zg;
    long long zs_partial[2], zs_dst[2];
};

// (bx / gx / by: the workgroup's position in the launch's own grid, or in its share of a batched launch — reduce_multi_kernel)
__device__ __forceinline__ void wreduce_body(const WreducePair& pair, int bx, int gx, int by) {
    const int e = by / pair.zg, zi = by % pair.zg;
    const WreduceParams& p = pair.r[e];
    const float* const partial_p = p.partial + (size_t)zi * pair.zs_partial[e];
    float* const dst_p = p.dst + (size_t)zi * pair.zs_dst[e];
    // one thread: 4 consecutive a of one (tap, g); the splits in four interleaved running sums, combined in a fixed order
    const int a4n = p.apad >> 2;
    const int total4 = p.ntaps * p.gpad * a4n;
    const size_t total = (size_t)total4 * 4;
    for (int i = bx * 256 + threadIdx.x; i < total4; i += gx * 256) {
        const int a = (i % a4n) * 4;
        const int r = i / a4n;
        const int g = r % p.gpad, tap = r / p.gpad;
        if (a >= p.cin) continue;
        int co = g, k = tap;
        if (p.transposed) {
            const int ph = g / p.cout_pad;
            co = g - ph * p.cout_pad;
            if (ph >= p.n_phase) continue;
            k = p.tap_k[ph][tap];
            if (k < 0) continue;
        }
        if (co >= p.cout) continue;
        if (p.gemm_cin > 0 && a >= p.gemm_cin * p.K) continue;
        const float4* src = reinterpret_cast<const float4*>(partial_p) + i;
        const size_t stride = total / 4;
        float4 s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        int sp = 0;
        for (; sp + 4 <= p.nsplit; sp += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = src[(size_t)(sp + j) * stride];
                s[j].x += v.x, s[j].y += v.y, s[j].z += v.z, s[j].w += v.w;
            }
        }
        for (int j = 0; sp < p.nsplit; ++sp, ++j) {
            const float4 v = src[(size_t)sp * stride];
            s[j].x += v.x, s[j].y += v.y, s[j].z += v.z, s[j].w += v.w;
        }
        const float o[4] = {(s[0].x + s[1].x) + (s[2].x + s[3].x), (s[0].y + s[1].y) + (s[2].y + s[3].y), (s[0].z + s[1].z) + (s[2].z + s[3].z),
                            (s[0].w + s[1].w) + (s[2].w + s[3].w)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (a + j >= p.cin) break;
            size_t d = p.transposed ? ((size_t)(a + j) * p.cout + co) * p.K + k : ((size_t)co * p.cin + a + j) * p.K + k;
            if (p.gemm_cin > 0) {
                const int tap = (a + j) / p.gemm_cin, c = (a + j) - tap * p.gemm_cin;
                if (tap >= p.K) break;
                d = ((size_t)co * p.gemm_cin + c) * p.K + tap;
            }
            if (p.poly_s > 0) {
                const int rr = (a + j) / p.poly_cin, c = (a + j) - rr * p.poly_cin;
                const int kk = p.poly_s * k + rr;
                if (rr >= p.poly_s || kk >= p.K) continue;
                d = ((size_t)co * p.poly_cin + c) * p.K + kk;
            }
            dst_p[d] = p.accumulate ? dst_p[d] + o[j] : o[j];
        }
    }
}

__global__ __launch_bounds__(256) void wreduce_kernel(const WreducePair pair) { wreduce_body(pair, blockIdx.x, gridDim.x, blockIdx.y); }

// The same reduction for the GEMM form of a conv with KT taps (discriminators: partial [split][g][a = tap * gemm_cin + c], gemm_cin % 4 == 0):
// a thread owns 4 consecutive c of one g for ALL taps, so its 4 * KT results are CONTIGUOUS in dst[(co * gemm_cin + c) * KT + tap] — whole
// 16-byte stores, a wave writes one contiguous run — where the generic kernel above scatters 4-byte stores KT floats apart (measured 1.8 TB/s
// on the 1024 x 5120 layers, which this form is for).  Same summation order per element as the generic kernel: bit-identical results.
template <int KT>
__device__ __forceinline__ void wreduce_gemm_body(const WreducePair& pair, int bx, int gx, int by) {
    const int zi = by % pair.zg;  // (one entry per launch in this form)
    const WreduceParams& p = pair.r[0];
    const float* const partial_p = p.partial + (size_t)zi * pair.zs_partial[0];
    float* const dst_p = p.dst + (size_t)zi * pair.zs_dst[0];
    const int c4n = p.gemm_cin >> 2;
    const int total = p.cout * c4n;
    const size_t stride4 = ((size_t)p.gpad * p.apad) >> 2;  // one split's partial, in float4
    for (int i = bx * 256 + threadIdx.x; i < total; i += gx * 256) {
        const int c4 = (i % c4n) * 4, co = i / c4n;
        const float4* src = reinterpret_cast<const float4*>(partial_p + (size_t)co * p.apad + c4);
        float o[4][KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const float4* st = src + ((t * p.gemm_cin) >> 2);
            float4 s[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            int sp = 0;
            for (; sp + 4 <= p.nsplit; sp += 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 v = st[(size_t)(sp + j) * stride4];
                    s[j].x += v.x, s[j].y += v.y, s[j].z += v.z, s[j].w += v.w;
                }
            }
            for (int j = 0; sp < p.nsplit; ++sp, ++j) {
                const float4 v = st[(size_t)sp * stride4];
                s[j].x += v.x, s[j].y += v.y, s[j].z += v.z, s[j].w += v.w;
            }
            o[0][t] = (s[0].x + s[1].x) + (s[2].x + s[3].x);
            o[1][t] = (s[0].y + s[1].y) + (s[2].y + s[3].y);
            o[2][t] = (s[0].z + s[1].z) + (s[2].z + s[3].z);
            o[3][t] = (s[0].w + s[1].w) + (s[2].w + s[3].w);
        }
        float4* d4 = reinterpret_cast<float4*>(dst_p + ((size_t)co * p.gemm_cin + c4) * KT);  // 16 * KT-byte aligned
        const float* of = &o[0][0];
#pragma unroll
        for (int q = 0; q < KT; ++q) {
            float4 v = make_float4(of[4 * q], of[4 * q + 1], of[4 * q + 2], of[4 * q + 3]);
            if (p.accumulate) {
                const float4 u = d4[q];
                v.x = u.x + v.x, v.y = u.y + v.y, v.z = u.z + v.z, v.w = u.w + v.w;
            }
            d4[q] = v;
        }
    }
}

template <int KT>
__global__ __launch_bounds__(256) void wreduce_gemm_kernel(const WreducePair pair) {
    wreduce_gemm_body<KT>(pair, blockIdx.x, gridDim.x, blockIdx.y);
}

// db[co] = sum over splits (and over the phases of a ConvTranspose1d) of the column sums
struct BreduceParams {
    const float* partial;
    float* dst;
    int nsplit, pitch, cout, cout_pad, n_phase;
    int accumulate;  // as WreduceParams::accumulate
};

// 16 channels x 16 strands per workgroup: strand j sums terms j, j + 16, ... (four loads in flight), the strands are added in a fixed order
struct BreducePair {
    BreduceParams b[2];
    int zg;
    long long zs_partial[2], zs_dst[2];
};

__device__ __forceinline__ void breduce_body(const BreducePair& pair, int bx, int by) {
    BreduceParams p = pair.b[by / pair.zg];
    if (!p.dst) return;
    {
        const int e = by / pair.zg, zi = by % pair.zg;
        p.partial += (size_t)zi * pair.zs_partial[e];
        p.dst += (size_t)zi * pair.zs_dst[e];
    }
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, jl = threadIdx.x >> 4;
    const int co = bx * 16 + cl;
    const int n = p.n_phase * p.nsplit;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (co < p.cout) {
        auto term = [&](int j) -> float {
            if (j >= n) return 0.f;
            const int ph = j / p.nsplit, sp = j - ph * p.nsplit;
            return p.partial[(size_t)sp * p.pitch + ph * p.cout_pad + co];
        };
        for (int j = jl; j < n; j += 64) {
            const float v0 = term(j), v1 = term(j + 16), v2 = term(j + 32), v3 = term(j + 48);
            s0 += v0, s1 += v1, s2 += v2, s3 += v3;
        }
    }
    red[jl][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (jl == 0 && co < p.cout) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += red[j][cl];
        p.dst[co] = p.accumulate ? p.dst[co] + s : s;
    }
}

__global__ __launch_bounds__(256) void breduce_kernel(const BreducePair pair) { breduce_body(pair, blockIdx.x, blockIdx.y); }

// Every pending reduction of a backward pass (or of one gradient bucket of it) in ONE launch: the weight-gradient kernels of consecutive layers
// then follow each other without a reduction — a launch that reads tens of MB with a handful of workgroups' worth of parallelism per layer —
// in between, and the reductions of all layers share the chip.  The table lives in device memory; weight job j owns workgroups
// wstart[j] .. wstart[j + 1] - 1 as a (wmeta[2 j] x rest) grid of form wmeta[2 j + 1] (0 generic, 3 / 5: GEMM form with that many taps), the
// bias jobs follow from workgroup wstart[nw] on.
__device__ __forceinline__ int find_job(const int* start, int njobs, int wg);  // (below)

struct ReduceTable {
    const WreducePair* w;
    const int* wstart;
    const int* wmeta;
    int nw;
    const BreducePair* b;
    const int* bstart;
    const int* bmeta;
    int nb;
};

__global__ __launch_bounds__(256) void reduce_multi_kernel(const ReduceTable t) {
    int wg = blockIdx.x;
    const int w_total = t.wstart[t.nw];
    if (wg < w_total) {
        const int j = find_job(t.wstart, t.nw, wg);
        const int rel = wg - t.wstart[j], gx = t.wmeta[2 * j], kind = t.wmeta[2 * j + 1];
        const int bx = rel % gx, by = rel / gx;
        if (kind == 5) wreduce_gemm_body<5>(t.w[j], bx, gx, by);
        else if (kind == 3) wreduce_gemm_body<3>(t.w[j], bx, gx, by);
        else wreduce_body(t.w[j], bx, gx, by);
        return;
    }
    wg -= w_total;
    const int j = find_job(t.bstart, t.nb, wg);
    const int rel = wg - t.bstart[j], gx = t.bmeta[j];
    breduce_body(t.b[j], rel % gx, rel / gx);
}

// ------------------------------------------------------------------------------------------------
// Output conv + tanh backward (hifigan.py:146-159, 231).  Forward: s = bias + sum_k sum_c lrelu_0.01(m[t + k - pad, c]) * w[k][c],
// out = tanh(s), m = MRF mean of the last stage.  ds = dout * (1 - out^2).
//   outconv_bwd_data   : dm[t, c] = lrelu'(m[t, c]) * sum_k ds[t - k + pad] * w[k][c] / nin   (gradient of EACH ResBlock output)
//   outconv_bwd_weight : partial[wg][k * C + c] = sum_t ds[t] * lrelu(m[t + k - pad, c]), partial[wg][K * C] = sum_t ds[t]
// ------------------------------------------------------------------------------------------------
struct OutBwdParams {
    const float* x0;
    const float* x1;
    const float* x2;
    const float* x3;
    int nin;
    const float* w;     // [k][Cp]
    const float* dout;  // (B, L) gradient of the waveform, row pitch dout_bstride
    const float* out;   // (B, L) forward waveform (for tanh')
    int64_t io_bstride;
    float* dm;          // (B, L, Cp): gradient of each ResBlock output of the last stage (the same for all nin)
    float* partial;     // [gridDim.x * gridDim.y][K * Cp + 1]
    int L, Cp, K;
    float slope;
    int use_tanh;
};

__device__ __forceinline__ float mrf_mean_at(const OutBwdParams& p, size_t off) {
    float v = p.x0[off];
    if (p.nin == 2) v = (v + p.x1[off]) / 2.0f;
    else if (p.nin == 3) v = ((v + p.x1[off]) + p.x2[off]) / 3.0f;
    else if (p.nin == 4) v = (((v + p.x1[off]) + p.x2[off]) + p.x3[off]) / 4.0f;
    return v;
}

__global__ __launch_bounds__(256) void outconv_bwd_data_kernel(const OutBwdParams p) {
    // one thread = one (row, 4 channels); ds of the K neighbouring samples recomputed on the fly (K = 7 scalar loads, cached)
    const int c4n = p.Cp >> 2;
    const long long total = (long long)p.L * c4n;
    const int seq = blockIdx.y;
    const int pad = (p.K - 1) / 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int t = (int)(i / c4n), c = (int)(i - (long long)t * c4n) * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < p.K; ++k) {
            const int ts = t - k + pad;
            if (ts < 0 || ts >= p.L) continue;
            const float o = p.out[(size_t)seq * p.io_bstride + ts];
            const float ds = p.dout[(size_t)seq * p.io_bstride + ts] * (p.use_tanh ? 1.f - o * o : 1.f);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(p.w + (size_t)k * p.Cp + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(ds, wv[e], acc[e]);
        }
        const size_t off = ((size_t)seq * p.L + t) * p.Cp + c;
        f32x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float m = mrf_mean_at(p, off + e);
            o4[e] = acc[e] * (m > 0.f ? 1.f : p.slope) / (float)p.nin;
        }
        *reinterpret_cast<f32x4*>(p.dm + off) = o4;
    }
}

__global__ __launch_bounds__(256) void outconv_bwd_weight_kernel(const OutBwdParams p) {
    // workgroup (x, seq) owns rows [x * rows_per, ...); thread = one (k, c) weight (K * Cp <= 256 * n) accumulates over the rows
    extern __shared__ float sds[];  // ds of the workgroup's rows
    const int seq = blockIdx.y;
    const int rows_per = (p.L + gridDim.x - 1) / gridDim.x;
    const int t0 = blockIdx.x * rows_per, t1 = min(t0 + rows_per, p.L);
    const int pad = (p.K - 1) / 2;
    for (int t = t0 + threadIdx.x; t < t1; t += 256) {
        const float o = p.out[(size_t)seq * p.io_bstride + t];
        sds[t - t0] = p.dout[(size_t)seq * p.io_bstride + t] * (p.use_tanh ? 1.f - o * o : 1.f);
    }
    __syncthreads();
    float* dst = p.partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (p.K * p.Cp + 1);
    for (int wi = threadIdx.x; wi < p.K * p.Cp; wi += 256) {
        const int k = wi / p.Cp, c = wi - k * p.Cp;
        float s = 0.f;
        for (int t = t0; t < t1; ++t) {
            const int ta = t + k - pad;
            if (ta < 0 || ta >= p.L) continue;
            const float m = mrf_mean_at(p, ((size_t)seq * p.L + ta) * p.Cp + c);
            s = fmaf(sds[t - t0], m > 0.f ? m : m * p.slope, s);
        }
        dst[wi] = s;
    }
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int t = t0; t < t1; ++t) s += sds[t - t0];
        dst[p.K * p.Cp] = s;
    }
}

// generic fixed-order sum of n partial vectors: dst[i] = sum_j partial[j * stride + i]  (+ optional (k, c) -> (c, k) transpose for the
// output conv weight, whose reference layout is (1, C, K))
struct VreduceParams {
    const float* partial;
    float* dst;
    int n, stride, len;
    int tr_K, tr_C, tr_Cp;  // tr_K > 0: i = k * Cp + c -> dst[c * K + k] for c < C
};

// (16 elements x 16 strands per workgroup, like breduce_kernel)
__global__ __launch_bounds__(256) void vreduce_kernel(const VreduceParams p) {
    __shared__ float red[16][17];
    const int il = threadIdx.x & 15, jl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + il;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < p.len) {
        const float* src = p.partial + i;
        int j = jl;
        for (; j + 48 < p.n; j += 64) {
            const float v0 = src[(size_t)j * p.stride], v1 = src[(size_t)(j + 16) * p.stride], v2 = src[(size_t)(j + 32) * p.stride],
                        v3 = src[(size_t)(j + 48) * p.stride];
            s0 += v0, s1 += v1, s2 += v2, s3 += v3;
        }
        for (; j < p.n; j += 16) s0 += src[(size_t)j * p.stride];
    }
    red[jl][il] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (jl != 0 || i >= p.len) return;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += red[j][il];
    if (p.tr_K > 0) {
        const int k = i / p.tr_Cp, c = i - k * p.tr_Cp;
        if (c < p.tr_C) p.dst[(size_t)c * p.tr_K + k] = s;
    } else {
        p.dst[i] = s;
    }
}

// out = ((a + b) + c) + d (the up to four ResBlock branches' gradients with respect to the upsample output)
__global__ __launch_bounds__(256) void add3_kernel(const float* a, const float* b, const float* c, const float* d, float* out, long long n4, int nin) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<const f32x4*>(a)[i];
        if (nin >= 2) {
            const f32x4 w = reinterpret_cast<const f32x4*>(b)[i];
            v = v + w;
        }
        if (nin >= 3) {
            const f32x4 w = reinterpret_cast<const f32x4*>(c)[i];
            v = v + w;
        }
        if (nin >= 4) {
            const f32x4 w = reinterpret_cast<const f32x4*>(d)[i];
            v = v + w;
        }
        reinterpret_cast<f32x4*>(out)[i] = v;
    }
}

// Gradient of the assembled input rows -> gradient of the features (B, cf, T) and of the AR features (B, ar_out) = sum over t
struct XinGradParams {
    const float* dxin;  // (B, T, cin_pad)
    float* dc;          // (B, cf, T) or null
    float* dfeat;       // (B, ar_out) or null
    int T, cf, ar_out, cin_pad;
};

__global__ __launch_bounds__(256) void xin_grad_kernel(const XinGradParams p) {
    const int b = blockIdx.x;
    const float* src = p.dxin + (size_t)b * p.T * p.cin_pad;
    if (p.dc)
        for (int i = threadIdx.x; i < p.cf * p.T; i += 256) {
            const int ch = i / p.T, t = i - ch * p.T;
            p.dc[(size_t)b * p.cf * p.T + i] = src[(size_t)t * p.cin_pad + ch];
        }
    if (p.dfeat)
        for (int j = threadIdx.x; j < p.ar_out; j += 256) {
            float s = 0.f;
            for (int t = 0; t < p.T; ++t) s += src[(size_t)t * p.cin_pad + p.cf + j];
            p.dfeat[(size_t)b * p.ar_out + j] = s;
        }
}

// ------------------------------------------------------------------------------------------------
// Conditioning branches, backward (hifigan.py:176-189, 212-220, 232-237).
//   speaker:  c = c + spk_fc(spk_emb_mat[spk_id]).unsqueeze(2) on the first `ch` (= feature + AR) channels
//     cond_spk_colsum : dspk[b, ch] = sum_t dxin[b, t, ch]
//     cond_spk_bwd    : d spk_fc.bias[ch] = sum_b dspk;  d spk_fc.weight[ch, e] = sum_b dspk[b, ch] * emb[id[b], e];
//                       d spk_emb_mat[s, e] = sum_{b: id[b] == s} sum_ch dspk[b, ch] * W[ch, e]          (fixed summation orders)
//   phoneme:  channels [ph_off, ph_off + ph_e) of frame t hold ph_emb_mat[ph[b, t]]
//     cond_ph_emb_bwd : d ph_emb_mat[s, e] = sum_{(b, t): ph[b, t] == s} dxin[b, t, ph_off + e]
//   phoneme-loss head: ph_out[b, q, f] = (1 / 2hop) * sum_{t in window(f)} (ph_fc.weight[q, :] . m[b, t, :] + bias[q]), m = last stage's MRF mean,
//   window(f) = [f hop - hop / 2, f hop + 3 hop / 2) clipped to the sequence (AvgPool1d(2 hop, hop, hop / 2), zero padding counted)
//     ph_head_bwd     : per (f, b): g[q] = dph_out[b, q, f] / 2hop;  dwin[b, f, c] = sum_q W[q, c] g[q];
//                       partial[(b, f)][q * C + c] = g[q] * windowsum(m)[c];  partial[(b, f)][num_ph * C + q] = g[q] * |window|
//     ph_dm_add       : dm[b, t, c] += (dwin[b, f_hi, c] + dwin[b, f_hi - 1, c]) / nin,  f_hi = (t + hop / 2) / hop   (the two windows holding t)
// ------------------------------------------------------------------------------------------------
struct CondBwdParams {
    const float* dxin;  // (B, T, cin_pad)
    int B, T, cin_pad;
    const int* spk_id;  // (B)
    const float* spk_emb;  // (num_spk, spk_e)
    const float* spk_w;    // (ch, spk_e)
    int spk_e, ch, num_spk;
    float* dspk;        // (B, ch) scratch
    float* d_spk_emb;
    float* d_spk_w;
    float* d_spk_b;
    const int* ph;      // (B, T)
    int ph_e, ph_off, num_ph;
    float* d_ph_emb;    // (num_ph, ph_e)
};

__global__ __launch_bounds__(256) void cond_spk_colsum_kernel(const CondBwdParams p) {
    const int b = blockIdx.x;
    const float* src = p.dxin + (size_t)b * p.T * p.cin_pad;
    for (int ch = threadIdx.x; ch < p.ch; ch += 256) {
        float s = 0.f;
        for (int t = 0; t < p.T; ++t) s += src[(size_t)t * p.cin_pad + ch];
        p.dspk[(size_t)b * p.ch + ch] = s;
    }
}

__global__ __launch_bounds__(256) void cond_spk_bwd_kernel(const CondBwdParams p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < p.ch) {  // spk_fc: row `ch` of the weight and its bias element
        float sb = 0.f;
        for (int b = 0; b < p.B; ++b) sb += p.dspk[(size_t)b * p.ch + i];
        p.d_spk_b[i] = sb;
        for (int e = 0; e < p.spk_e; ++e) {
            float s = 0.f;
            for (int b = 0; b < p.B; ++b) s = fmaf(p.dspk[(size_t)b * p.ch + i], p.spk_emb[(size_t)p.spk_id[b] * p.spk_e + e], s);
            p.d_spk_w[(size_t)i * p.spk_e + e] = s;
        }
    }
    const int j = i - ((p.ch + 255) / 256) * 256;  // the workgroups after those: one thread per embedding element
    if (j >= 0 && j < p.num_spk * p.spk_e) {
        const int sidx = j / p.spk_e, e = j - sidx * p.spk_e;
        float s = 0.f;
        for (int b = 0; b < p.B; ++b) {
            if (p.spk_id[b] != sidx) continue;
            float sv = 0.f;
            for (int ch = 0; ch < p.ch; ++ch) sv = fmaf(p.dspk[(size_t)b * p.ch + ch], p.spk_w[(size_t)ch * p.spk_e + e], sv);
            s += sv;
        }
        p.d_spk_emb[j] = s;
    }
}

__global__ __launch_bounds__(256) void cond_ph_emb_bwd_kernel(const CondBwdParams p) {
    // workgroup = one phoneme; thread (strand, e): strands split the (b, t) positions, combined in a fixed order
    __shared__ float red[256];
    const int sidx = blockIdx.x;
    const int e = threadIdx.x % p.ph_e, strand = threadIdx.x / p.ph_e, nstrands = 256 / p.ph_e;
    float s = 0.f;
    if (strand < nstrands)
        for (int i = strand; i < p.B * p.T; i += nstrands)
            if (p.ph[i] == sidx) s += p.dxin[(size_t)i * p.cin_pad + p.ph_off + e];
    red[threadIdx.x] = strand < nstrands ? s : 0.f;
    __syncthreads();
    if (threadIdx.x < p.ph_e) {
        float t = 0.f;
        for (int q = 0; q < nstrands; ++q) t += red[q * p.ph_e + threadIdx.x];
        p.d_ph_emb[(size_t)sidx * p.ph_e + threadIdx.x] = t;
    }
}

struct PhHeadBwdParams {
    const float* x0;
    const float* x1;
    const float* x2;
    const float* x3;
    int nin;
    const float* w;        // ph_fc.weight (num_ph, C)
    const float* dph_out;  // (B, num_ph, T)
    float* dwin;           // (B, T, Cp)
    float* partial;        // [B * T][num_ph * (C + 1)]
    float* dm;             // (B, L, Cp): accumulated into
    int C, Cp, L, T, hop, num_ph;
};

__global__ __launch_bounds__(256) void ph_head_bwd_kernel(const PhHeadBwdParams p) {
    __shared__ float part[8][128];
    __shared__ float wsum[128];
    __shared__ float g[256];
    const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int t0 = f * p.hop - p.hop / 2, K = 2 * p.hop;
    const int lo = max(t0, 0), hi = min(t0 + K, p.L);
    const int ch = tid & 31, rs = tid >> 5;
    for (int c0 = 0; c0 < p.C; c0 += 32) {  // the window's sum of the MRF mean, summed exactly as ph_head_kernel does
        float s = 0.f;
        if (c0 + ch < p.C)
            for (int t = lo + rs; t < hi; t += 8) {
                const size_t off = ((size_t)b * p.L + t) * p.Cp + c0 + ch;
                float v = p.x0[off];
                if (p.nin == 2) v = (v + p.x1[off]) / 2.0f;
                else if (p.nin == 3) v = ((v + p.x1[off]) + p.x2[off]) / 3.0f;
                else if (p.nin == 4) v = (((v + p.x1[off]) + p.x2[off]) + p.x3[off]) / 4.0f;
                s += v;
            }
        part[rs][c0 + ch] = s;
    }
    for (int q = tid; q < p.num_ph; q += 256) g[q] = p.dph_out[((size_t)b * p.num_ph + q) * p.T + f] / (float)K;
    __syncthreads();
    for (int c = tid; c < p.C; c += 256) {
        float s = 0.f;
        for (int r = 0; r < 8; ++r) s += part[r][c];
        wsum[c] = s;
    }
    __syncthreads();
    for (int c = tid; c < p.Cp; c += 256) {
        float s = 0.f;
        if (c < p.C)
            for (int q = 0; q < p.num_ph; ++q) s = fmaf(p.w[(size_t)q * p.C + c], g[q], s);
        p.dwin[((size_t)b * p.T + f) * p.Cp + c] = s;
    }
    float* dst = p.partial + ((size_t)b * p.T + f) * ((size_t)p.num_ph * (p.C + 1));
    for (int i = tid; i < p.num_ph * p.C; i += 256) {
        const int q = i / p.C, c = i - q * p.C;
        dst[i] = g[q] * wsum[c];
    }
    for (int q = tid; q < p.num_ph; q += 256) dst[p.num_ph * p.C + q] = g[q] * (float)(hi - lo);
}

__global__ __launch_bounds__(256) void ph_dm_add_kernel(const PhHeadBwdParams p) {
    const int c4n = p.Cp >> 2;
    const long long total = (long long)p.L * c4n;
    const int b = blockIdx.y;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int t = (int)(i / c4n), c = (int)(i - (long long)t * c4n) * 4;
        const int fh = (t + p.hop / 2) / p.hop;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (fh - 1 >= 0 && fh - 1 < p.T) a = *reinterpret_cast<const f32x4*>(p.dwin + ((size_t)b * p.T + fh - 1) * p.Cp + c);
        if (fh < p.T) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.dwin + ((size_t)b * p.T + fh) * p.Cp + c);
            a = a + v;
        }
        f32x4* d = reinterpret_cast<f32x4*>(p.dm + ((size_t)b * p.L + t) * p.Cp + c);
        f32x4 o = *d;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += a[e] / (float)p.nin;
        *d = o;
    }
}

// ------------------------------------------------------------------------------------------------
// PastFCEncoder backward (pytorch_layers.py:438-460): h0 = ar, h_{l+1} = lrelu_0.1(W_l h_l + b_l) for l < 4, feat = W_4 h_4 + b_4.
//   mlp_bwd_delta : one workgroup per utterance: delta_4 = dfeat; delta_{l-1} = (W_l^T delta_l) * lrelu'(h_l); also dar = W_0^T delta_0
//   mlp_bwd_weight: dW_l[o, i] = sum_b delta_l[b, o] * h_l[b, i];  db_l[o] = sum_b delta_l[b, o]
// h_l (l = 0..4; h_0 = the AR context) come from the forward's tape: (B, 5, 1024) floats; deltas go to (B, 5, 512).
// ------------------------------------------------------------------------------------------------
struct MlpBwdParams {
    const float* h;      // (B, 5, 1024): inputs of the five Linear layers
    const float* dfeat;  // (B, ar_out)
    float* delta;        // (B, 5, 512): gradient of each Linear's output
    float* dar;          // (B, ar_input) or null
    const float* wt[5];  // transposed weights (in, out), as the forward uses them
    int dims[6];         // ar_input, hidden x 4, ar_output
    float* dW[5];        // reference layout (out, in)
    float* db[5];
    int B;
};

__global__ __launch_bounds__(256) void mlp_bwd_delta_kernel(const MlpBwdParams p) {
    __shared__ float d[2][512];
    const int b = blockIdx.x, tid = threadIdx.x;
    int cur = 0;
    for (int j = tid; j < p.dims[5]; j += 256) {
        const float v = p.dfeat[(size_t)b * p.dims[5] + j];
        d[0][j] = v;
        p.delta[((size_t)b * 5 + 4) * 512 + j] = v;
    }
    __syncthreads();
    for (int l = 4; l >= 0; --l) {
        const int din = p.dims[l], dout = p.dims[l + 1];
        const float* wt = p.wt[l];  // (din, dout)
        const float* hl = p.h + ((size_t)b * 5 + l) * 1024;
        for (int i = tid; i < din; i += 256) {
            float s = 0.f;
            for (int o = 0; o < dout; ++o) s = fmaf(wt[(size_t)i * dout + o], d[cur][o], s);
            if (l > 0) {
                s *= hl[i] > 0.f ? 1.f : 0.1f;  // h_l = lrelu(pre): same sign as the pre-activation
                d[cur ^ 1][i] = s;
                p.delta[((size_t)b * 5 + (l - 1)) * 512 + i] = s;
            } else if (p.dar) {
                p.dar[(size_t)b * din + i] = s;
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}

__global__ __launch_bounds__(256) void mlp_bwd_weight_kernel(const MlpBwdParams p) {
    const int l = blockIdx.y;
    const int din = p.dims[l], dout = p.dims[l + 1];
    const int n = din * dout;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n + dout; i += gridDim.x * 256) {
        float s = 0.f;
        if (i < n) {
            const int o = i / din, ii = i - o * din;
            for (int b = 0; b < p.B; ++b) s = fmaf(p.delta[((size_t)b * 5 + l) * 512 + o], p.h[((size_t)b * 5 + l) * 1024 + ii], s);
            p.dW[l][i] = s;
        } else {
            const int o = i - n;
            for (int b = 0; b < p.B; ++b) s += p.delta[((size_t)b * 5 + l) * 512 + o];
            p.db[l][o] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Device-side weight packing (training: weights live on the device and change every step).  One thread per packed fp32 element of the
// [n_block32][chunk][tap][c16][half][lane][4] fragment layout (see pack_w32 in hificar.hip), reading the reference-layout source.
//   mode 0  Conv1d forward            src (cout, cin, K)   w'[co][ci][k] = W[co][ci][k]
//   mode 1  ConvTranspose1d forward   src (cin, cout, K)   phase-major virtual output channels, taps through tap_k
//   mode 2  Conv1d data gradient      a Conv1d (cin -> its input channels): w'[co' = ci][ci' = co][k'] = W[co][ci][K - 1 - k']
//   mode 3  ConvTranspose1d data gradient as a Conv1d over the phase-major virtual channels (r, co) of dy:
//           w'[co' = ci][ci' = r * cout_pad + co][j] = scale * W[ci][co][r + s * (jmin + j) + pad] when that tap exists
// ------------------------------------------------------------------------------------------------
struct PackParams {
    const float* src;
    float* dst;
    long long total;   // packed elements
    int mode;
    int n_blocks32, nchunk, ntaps, nc16, chunk;
    int nb32_per_phase;
    int cin, cout, K;            // of the SOURCE weight (reference layout)
    int cin_pack, cout_pack;     // real (unpadded) input / output channels of the packed layer
    int stride, pad, cout_pad, jmin;
    float scale;
    int tap_k[kMaxPhase][kMaxTaps];
};

// modes 4-7: the parameters that are not conv weight packs, so that ONE batched launch re-sends every parameter
//   mode 4  output conv weight (1, C, K) -> [k][Cp]        cin = C, cout_pad = Cp
//   mode 5  transpose (rows = cout, cols = cin) -> (cols, rows)   (PastFCEncoder weights)
//   mode 6  bias replicated over the phases: dst[r * cout_pad + co] = src[co], total = n_phase * cout
//   mode 7  plain copy
//   modes 8 / 9  GEMM form of a (grouped, strided) conv and of its data gradient, src (cout_g, cin_g, k) of one group: the im2col
//           column j = tap * cin_g + c is the packed layer's input channel (8) / output channel (9)   (hificar_disc_kernels.hip.h)
//   modes 10 / 11  polyphase-input form of a strided conv, src (cout_g, cin_g = p.cin, k = p.K), stride p.stride: rows of `stride` input
//           positions are the packed layer's channels r * cin_g + c, tap q holds kernel index stride * q + r (zero past k).
//           10: the forward conv (cin' = stride * cin_g -> cout_g, taps 0 .. ntaps - 1);  11: its data gradient (cout_g -> cin', the taps
//           reversed: a Conv1d with padding ntaps - 1)
// (Index type I: int for packs below 2^30 elements — every one in practice — so that the lane / slab / tap / chunk decode runs on 32-bit
// divisions instead of 64-bit ones; long long otherwise.  Pure copies: the same packed tensor either way.)
template <typename I>
__device__ __forceinline__ void pack_w32_body_t(const PackParams& p, I first, I step) {
    const I total = (I)p.total;
    // Taps inside the thread (round 6).  The generic loop below reads ONE float per packed element, and the K taps of a (co, ci) pair — K adjacent floats
    // of the source — belong to K packed blocks thousands of elements apart: other workgroups, other XCDs, so every source sector was fetched by up to
    // K (x 8 L2s) readers (r06a: 1.58 GB moved per launch for 161 MB).  Here a thread walks the packed layout WITHOUT its tap axis, reads its pair's K
    // adjacent floats once and writes the K packed elements (each store still one contiguous run per wave).  Pure copies: the same packed tensor.
    //   modes 0 / 2 (Conv1d forward / data gradient) and 10 / 11 (polyphase-input forms: a thread's taps are every stride-th float of its pair's K):
    //   the tap axis of the packed layout;  modes 8 / 9 (GEMM forms): the taps are groups of cin / chunk K-chunks (8) resp. cin / 32 channel blocks
    //   (9) — when those divide evenly; everything else (transposed convs, odd shapes) takes the generic loop.
    {
        const bool one_phase = p.nb32_per_phase == p.n_blocks32;
        const bool conv_in = (p.mode == 0 || p.mode == 2) && p.ntaps == p.K && p.K > 1 && one_phase;
        const bool poly_in = (p.mode == 10 || p.mode == 11) && p.ntaps > 1 && one_phase && p.stride >= 1;
        const bool gemm_f = p.mode == 8 && p.ntaps == 1 && p.K > 1 && one_phase && p.cin % p.chunk == 0 && p.nchunk * p.chunk == p.K * p.cin &&
                            p.cin_pack == p.K * p.cin;
        const bool gemm_d = p.mode == 9 && p.ntaps == 1 && p.K > 1 && one_phase && p.cin % 32 == 0 && p.n_blocks32 * 32 == p.K * p.cin &&
                            p.cout_pack == p.K * p.cin;
        if (conv_in || poly_in || gemm_f || gemm_d) {
            const int K = p.K;
            const int T = poly_in ? p.ntaps : K;  // taps the thread writes
            const int nchunk_i = gemm_f ? p.cin / p.chunk : p.nchunk;  // chunks / channel blocks the thread index walks (a tap's share in the GEMM forms)
            const int nblk_i = gemm_d ? p.cin / 32 : p.n_blocks32;
            const I sub_total = (I)nblk_i * nchunk_i * p.nc16 * 512;
            for (I i = first; i < sub_total; i += step) {
                I r = i;
                const int j = (int)(r & 3);
                r >>= 2;
                const int lane = (int)(r & 63);
                r >>= 6;
                const int v = (int)(r & 1);
                r >>= 1;
                const int u = (int)(r % p.nc16);
                r /= p.nc16;
                const int c = (int)(r % nchunk_i);
                const int nb = (int)(r / nchunk_i);
                const int g = lane >> 5, n = lane & 31;
                const int co = nb * 32 + n;                          // (within a tap's share for mode 9)
                const int ci = c * p.chunk + u * 16 + 8 * v + 4 * g + j;  // (within a tap's share for mode 8)
                const int in_lane = ((u * 2 + v) * 64 + lane) * 4 + j;   // offset inside a (block, chunk, tap) fragment group of nc16 * 512 floats
                const float* sp = nullptr;
                int rr = 0;  // polyphase forms: this thread's position inside a row of `stride` inputs
                if (conv_in) {
                    if (co < p.cout_pack && ci < p.cin_pack) sp = p.mode == 0 ? p.src + ((size_t)co * p.cin + ci) * K : p.src + ((size_t)ci * p.cin + co) * K;
                } else if (poly_in) {
                    if (co < p.cout_pack && ci < p.cin_pack) {
                        if (p.mode == 10) {
                            rr = ci / p.cin;
                            sp = p.src + ((size_t)co * p.cin + (ci - rr * p.cin)) * K;
                        } else {
                            rr = co / p.cin;
                            sp = p.src + ((size_t)ci * p.cin + (co - rr * p.cin)) * K;
                        }
                    }
                } else if (gemm_f) {
                    if (co < p.cout_pack) sp = p.src + ((size_t)co * p.cin + ci) * K;  // ci < cin: the tap's own channel
                } else {
                    if (ci < p.cin_pack) sp = p.src + ((size_t)ci * p.cin + co) * K;   // co < cin
                }
                for (int t0 = 0; t0 < T; t0 += kMaxTaps) {  // (the scale discriminators' k = 41: three groups of taps)
                    float w[kMaxTaps];  // w[q]: the value of packed tap t0 + q (mode 2 packs the taps reversed, 10 / 11 every stride-th kernel index)
#pragma unroll
                    for (int q = 0; q < kMaxTaps; ++q) {
                        const int t = t0 + q;
                        int k = t;
                        if (p.mode == 2) k = K - 1 - t;
                        else if (p.mode == 10) k = p.stride * t + rr;
                        else if (p.mode == 11) k = p.stride * (T - 1 - t) + rr;
                        w[q] = (sp && t < T && k >= 0 && k < K) ? sp[k] : 0.f;
                    }
#pragma unroll
                    for (int q = 0; q < kMaxTaps; ++q) {
                        const int t = t0 + q;
                        if (t < T) {
                            size_t d;
                            if (conv_in || poly_in) d = (((size_t)nb * p.nchunk + c) * T + t) * p.nc16 * 512 + in_lane;
                            else if (gemm_f) d = ((size_t)nb * p.nchunk + (size_t)t * nchunk_i + c) * p.nc16 * 512 + in_lane;
                            else d = (((size_t)t * nblk_i + nb) * p.nchunk + c) * p.nc16 * 512 + in_lane;
                            p.dst[d] = w[q];
                        }
                    }
                }
            }
            return;
        }
    }
    for (I i = first; i < total; i += step) {
        if (p.mode >= 4 && p.mode <= 7) {
            if (p.mode == 4) {
                const int k = (int)(i / p.cout_pad), c = (int)(i - (long long)k * p.cout_pad);
                p.dst[i] = c < p.cin ? p.src[(size_t)c * p.K + k] : 0.f;
            } else if (p.mode == 5) {
                const int r = (int)(i / p.cin), c = (int)(i - (long long)r * p.cin);
                p.dst[(size_t)c * p.cout + r] = p.src[i];
            } else if (p.mode == 6) {
                const int r = (int)(i / p.cout), co = (int)(i - (long long)r * p.cout);
                p.dst[(size_t)r * p.cout_pad + co] = p.src[co];
            } else {
                p.dst[i] = p.src[i];
            }
            continue;
        }
        I r = i;
        const int j = (int)(r & 3);
        r >>= 2;
        const int lane = (int)(r & 63);
        r >>= 6;
        const int v = (int)(r & 1);
        r >>= 1;
        const int u = (int)(r % p.nc16);
        r /= p.nc16;
        const int t = (int)(r % p.ntaps);
        r /= p.ntaps;
        const int c = (int)(r % p.nchunk);
        const int nb = (int)(r / p.nchunk);
        float val = 0.f;
        if (nb < p.n_blocks32) {
            const int g = lane >> 5, n = lane & 31;
            const int phase = nb / p.nb32_per_phase;
            const int co = (nb % p.nb32_per_phase) * 32 + n;     // output channel of the packed layer (within its phase)
            const int ci = c * p.chunk + u * 16 + 8 * v + 4 * g + j;  // input channel of the packed layer
            if (co < p.cout_pack && ci < p.cin_pack) {
                if (p.mode == 0) {
                    val = p.src[((size_t)co * p.cin + ci) * p.K + t];
                } else if (p.mode == 1) {
                    const int k = p.tap_k[phase][t];
                    if (k >= 0) val = p.src[((size_t)ci * p.cout + co) * p.K + k];
                } else if (p.mode == 2) {
                    val = p.src[((size_t)ci * p.cin + co) * p.K + (p.K - 1 - t)];
                } else if (p.mode == 8) {  // GEMM form of a conv (discriminators): input column j = tap * cin + c
                    const int tap = ci / p.cin, c = ci - tap * p.cin;
                    val = p.src[((size_t)co * p.cin + c) * p.K + tap];
                } else if (p.mode == 9) {  // its data gradient: output column j = tap * cin + c, input channel = the conv's output channel
                    const int tap = co / p.cin, c = co - tap * p.cin;
                    val = p.src[((size_t)ci * p.cin + c) * p.K + tap];
                } else if (p.mode == 10) {
                    const int rr = ci / p.cin, c = ci - rr * p.cin;
                    const int kk = p.stride * t + rr;
                    if (kk < p.K) val = p.src[((size_t)co * p.cin + c) * p.K + kk];
                } else if (p.mode == 11) {
                    const int rr = co / p.cin, c = co - rr * p.cin;
                    const int kk = p.stride * (p.ntaps - 1 - t) + rr;
                    if (kk < p.K) val = p.src[((size_t)ci * p.cin + c) * p.K + kk];
                } else {
                    const int rr = ci / p.cout_pad, cc = ci - rr * p.cout_pad;  // virtual input channel -> (phase r, real co)
                    const int k = rr + p.stride * (p.jmin + t) + p.pad;
                    if (rr < p.stride && cc < p.cout && k >= 0 && k < p.K) val = p.scale * p.src[((size_t)co * p.cout + cc) * p.K + k];
                }
            }
        }
        p.dst[i] = val;
    }
}

__device__ __forceinline__ void pack_w32_body(const PackParams& p, long long first, long long step) {
    if (p.total < (1LL << 30) && step < (1LL << 30)) pack_w32_body_t<int>(p, (int)first, (int)step);
    else pack_w32_body_t<long long>(p, first, step);
}

__global__ __launch_bounds__(256) void pack_w32_kernel(const PackParams p) {
    pack_w32_body(p, (long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256);
}

// job of workgroup `wg` in a batched launch: start[j] <= wg < start[j + 1]
__device__ __forceinline__ int find_job(const int* start, int njobs, int wg) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (start[mid] <= wg) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// every pack of the generator in one launch (the table lives in device memory; job j owns workgroups start[j] .. start[j + 1] - 1)
__global__ __launch_bounds__(256) void pack_all_kernel(const PackParams* jobs, const int* start, int njobs) {
    const int j = find_job(start, njobs, blockIdx.x);
    const int lb = blockIdx.x - start[j], nb = start[j + 1] - start[j];
    pack_w32_body(jobs[j], (long long)lb * 256 + threadIdx.x, (long long)nb * 256);
}

// ------------------------------------------------------------------------------------------------
// Weight norm on the device (hifigan.py:268-278: torch.nn.utils.weight_norm, dim 0): w[r, :] = g[r] * v[r, :] / ||v[r, :]||.
//   param_gather_kernel  raw parameters -> the FOLDED master copy every pack reads (plain tensors are copied), one launch
//   wn_backward_kernel   folded gradients -> raw gradients: dg[r] = <dW[r], v[r]> / n,  dv[r] = (g / n) dW[r] - (g <dW[r], v[r]> / n^3) v[r]
// One workgroup per row (weight norm) or per 1024 floats (copies); sums in a fixed order.
// ------------------------------------------------------------------------------------------------
struct ParamJob {
    const float* v;   // weight_v, or the plain tensor
    const float* g;   // weight_g; nullptr: copy
    long long dst;    // float offset in the folded buffer (forward) / in the folded gradient buffer (backward: where dW lies)
    long long dv, dg; // backward: float offsets in the raw gradient buffer (copy: dv only)
    int rows, cols;   // weight norm: rows = dim 0, cols = the rest;  copy: cols = numel, rows = ceil(numel / 1024)
};

__device__ __forceinline__ float wg_sum256(float x, float* red) {  // all 256 threads get the total; fixed order
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void param_gather_kernel(const ParamJob* jobs, const int* start, int njobs, float* folded) {
    __shared__ float red[4];
    const int j = find_job(start, njobs, blockIdx.x);
    const ParamJob q = jobs[j];
    const int r = blockIdx.x - start[j];
    float* dst = folded + q.dst;
    if (!q.g) {
        const int i = r * 1024 + threadIdx.x;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * 256 < q.cols) dst[i + u * 256] = q.v[i + u * 256];
        return;
    }
    const float* v = q.v + (size_t)r * q.cols;
    float ss = 0.f;
    for (int c = threadIdx.x; c < q.cols; c += 256) ss += v[c] * v[c];
    ss = wg_sum256(ss, red);
    const float sc = q.g[r] / sqrtf(ss);
    for (int c = threadIdx.x; c < q.cols; c += 256) dst[(size_t)r * q.cols + c] = v[c] * sc;
}

// (wg0: first workgroup of the table this launch covers — a bucket of jobs launches start[lo] .. start[hi] - 1 only)
// scale: device scalar every raw gradient is multiplied by (the upstream gradient of a scalar loss), or null
__global__ __launch_bounds__(256) void wn_backward_kernel(const ParamJob* jobs, const int* start, int njobs, const float* grads, float* raw, int wg0,
                                                          const float* scale) {
    __shared__ float red[4];
    const float sc = scale ? *scale : 1.f;
    const int wg = (int)blockIdx.x + wg0;
    const int j = find_job(start, njobs, wg);
    const ParamJob q = jobs[j];
    const int r = wg - start[j];
    const float* dw = grads + q.dst;
    if (!q.g) {
        const int i = r * 1024 + threadIdx.x;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * 256 < q.cols) raw[q.dv + i + u * 256] = scale ? dw[i + u * 256] * sc : dw[i + u * 256];
        return;
    }
    const float* v = q.v + (size_t)r * q.cols;
    dw += (size_t)r * q.cols;
    float ss = 0.f, dot = 0.f;
    for (int c = threadIdx.x; c < q.cols; c += 256) {
        const float x = v[c];
        ss += x * x;
        dot += x * dw[c];
    }
    ss = wg_sum256(ss, red);
    dot = wg_sum256(dot, red);
    const float n = sqrtf(ss), g = q.g[r];
    const float a = g / n, b = g * dot / (n * ss);
    if (threadIdx.x == 0) raw[q.dg + r] = scale ? (dot / n) * sc : dot / n;
    float* dv = raw + q.dv + (size_t)r * q.cols;
    if (scale) {
        for (int c = threadIdx.x; c < q.cols; c += 256) dv[c] = (a * dw[c] - b * v[c]) * sc;
    } else {
        for (int c = threadIdx.x; c < q.cols; c += 256) dv[c] = a * dw[c] - b * v[c];
    }
}

}  // namespace hificar
