// HIP kernels for the HiFi-GAN / HiFi-CAR generator forward pass on gfx950 (CDNA4, wave64).
//
// Internal activation layout is CHANNELS-LAST: (sequence, time, channel) fp32 with the channel
// axis contiguous.  The reference's (B, C, T) layout exists only at the C-ABI boundary (feature
// input, waveform output), which `front_kernel` / `output_conv_kernel` convert on the fly.
// Channels-last makes every halo a whole-row affair (zero rows outside [0, L) reproduce the
// reference's per-conv zero padding exactly), keeps global loads 16-byte aligned for any tap
// offset, and puts the GEMM reduction axis (input channels) contiguous for MFMA operand reads.
//
// Every Conv1d and every ConvTranspose1d of the generator runs through ONE implicit-GEMM kernel
// family (`conv_mfma_f32_kernel`):  D[t, co] = sum_{tap} sum_{ci} act(X)[t + off(tap), ci] * W[tap][ci][co]
// with M = time, N = output channels, K = taps x input channels.  A ConvTranspose1d with K = 2*stride
// is the same contraction with N = stride*Cout "virtual" channels (phase-major) and per-phase tap
// lists, because out[(q*s + r), co] in channels-last memory IS row q, column r*Cout + co.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hificar {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kMaxPhase = 8;
constexpr int kMaxTaps = 16;

// The A operand (activations) is produced while staging into LDS from `nin` inputs:
//   nin = 1: x0;  nin = 2: (x0 + x1) / 2;  nin = 3: ((x0 + x1) + x2) / 3
// (the MRF mean `cs / num_blocks` of reference hifigan.py:226-230, summed in the reference's order).

struct ConvParams {
    const float* x0;
    const float* x1;
    const float* x2;
    const float* w;     // f32 path: packed [n_block][tap][ci][NB]
    const bf16x8* w16;  // bf16x3 path: packed MFMA B fragments [n_block32][chunk][tap][c16][hi|lo][lane], 16 B each
    const float* bias;  // [cout_total] (never null; zeros when the layer has no bias)
    const float* res;   // residual, same layout as y, or null
    float* y;
    int L;              // rows (time steps) per sequence, input rows == output rows
    int tiles_per_seq;  // ceil(L / TM)
    int cin;            // padded input channels == row pitch of x*
    int cout_total;     // row pitch of y / res / bias length
    int chunk;          // input-channel chunk staged per pass (multiple of 8, divides cin)
    int n_blocks;       // number of NB-wide output blocks (cout_total / NB)
    int nb_per_phase;   // n_blocks / n_phase
    int n_blocks32;     // bf16x3 path: number of 32-wide output blocks (cout_total / 32)
    int nb32_per_phase;
    int ntaps;          // taps per phase (same for all phases; missing taps have zero weights)
    int off_min;        // min over all tap offsets (<= 0)
    int halo;           // off_max - off_min
    int nin;            // number of inputs averaged while staging (1..3)
    float slope;        // LeakyReLU slope applied to the staged input; 1.0f = identity
    // tap t of phase r reads input row  t_out + tap_off0[r] + t * tap_step  (an arithmetic progression for both
    // Conv1d: -padding + t*dilation, and the polyphase ConvTranspose1d: floor((r+p)/s) - t); no per-tap table
    // lookups in the K loop (a memory lookup there would drain the weight prefetch queue with vmcnt(0)).
    int tap_step;
    int tap_off0[kMaxPhase];
};

struct MultiConvParams {
    ConvParams p[3];
};

__device__ __forceinline__ float lrelu(float v, float slope) { return v >= 0.f ? v : v * slope; }

// ------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate).
//
//   workgroup = 4 waves arranged WM (time) x WN (channel blocks); a wave owns MI x NJ MFMA tiles
//   of 32x32, i.e. MI*32 time rows x NB = NJ*32 output channels.  TM = WM*MI*32 rows per workgroup.
//   grid = (sequences * tiles_per_seq, ceil(n_blocks / WN), branches)
//
//   LDS holds act(X)[t0 + off_min .. t0 + TM + off_max) x chunk channels, row pitch chunk+4 floats
//   (pitch/4 odd => the ds_read_b128 of 16 lanes x distinct rows hit 16 distinct 16-byte slots).
//   Weights are NOT staged: each lane's B operand is a contiguous float<NJ> of the packed weight
//   row, 512 B per half-wave, L1/L2-served (the whole 54 MB model sits in the 256 MB Infinity Cache).
//
//   MFMA operand maps (32x32x2 f32): A lane l -> A[i = l&31][k = l>>5], B lane l -> B[k = l>>5][n = l&31],
//   D reg r -> D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].  K is consumed 8 channels at a time:
//   the half-wave g = l>>5 takes channels c8 + 4g .. c8 + 4g + 3 (one b128 LDS read = 4 MFMA steps).
//   Output column n of tile j is channel NJ*n + j, so a lane's NJ accumulators of one row are NJ
//   adjacent floats in memory (vector store).
// ------------------------------------------------------------------------------------------------
template <int MI, int NJ, int WM, int WN>
__global__ __launch_bounds__(256) void conv_mfma_f32_kernel(const MultiConvParams mp) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int NB = NJ * 32;
    constexpr int TM = WM * MI * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const ConvParams& p = mp.p[blockIdx.z];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int li = lane & 31;
    const int g = lane >> 5;

    const int seq = blockIdx.x / p.tiles_per_seq;
    const int t0 = (blockIdx.x % p.tiles_per_seq) * TM;
    const int nb = blockIdx.y * WN + wn;
    const bool active = nb < p.n_blocks;
    const int phase = active ? nb / p.nb_per_phase : 0;

    const int P = p.chunk + 4;            // LDS row pitch (floats)
    const int R = TM + p.halo;            // staged rows
    const int c4n = p.chunk >> 2;         // float4 per staged row
    const size_t seq_base = (size_t)seq * p.L;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][j][r] = 0.f;

    const float* wblk = p.w + (size_t)(active ? nb : 0) * p.ntaps * p.cin * NB;
    const int wave_row0 = wm * (MI * 32);
    const float slope = p.slope;
    const int roff0 = __builtin_amdgcn_readfirstlane(p.tap_off0[phase] - p.off_min);

    for (int c0 = 0; c0 < p.cin; c0 += p.chunk) {
        __syncthreads();
        // ---- stage act(X)[rows, c0 : c0+chunk] into LDS (zero rows outside the sequence) ----
        for (int idx = tid; idx < R * c4n; idx += 256) {
            const int r = idx / c4n;
            const int c4 = idx - r * c4n;
            const int t = t0 + p.off_min + r;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (t >= 0 && t < p.L) {
                const size_t off = (seq_base + t) * p.cin + c0 + c4 * 4;
                v = *reinterpret_cast<const f32x4*>(p.x0 + off);
                if (p.nin == 3) {
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(p.x1 + off);
                    const f32x4 v2 = *reinterpret_cast<const f32x4*>(p.x2 + off);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ((v[e] + v1[e]) + v2[e]) / 3.0f;
                } else if (p.nin == 2) {
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(p.x1 + off);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (v[e] + v1[e]) / 2.0f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = lrelu(v[e], slope);
            }
            *reinterpret_cast<f32x4*>(&smem[r * P + c4 * 4]) = v;
        }
        __syncthreads();
        if (!active) continue;
        // ---- MFMA over taps x channels of this chunk ----
        for (int t = 0; t < p.ntaps; ++t) {
            const int roff = roff0 + t * p.tap_step;  // >= 0
            const float* arow = &smem[(wave_row0 + li + roff) * P + 4 * g];
            const float* wrow = wblk + ((size_t)t * p.cin + c0 + 4 * g) * NB + NJ * li;
            for (int c8 = 0; c8 < p.chunk; c8 += 8) {
                f32x4 a[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(arow + mi * 32 * P + c8);
                float b[4][NJ];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float* wp = wrow + (size_t)(c8 + s) * NB;
                    if constexpr (NJ == 4) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(wp);
                        b[s][0] = v[0]; b[s][1] = v[1]; b[s][2] = v[2]; b[s][3] = v[3];
                    } else if constexpr (NJ == 2) {
                        const f32x2 v = *reinterpret_cast<const f32x2*>(wp);
                        b[s][0] = v[0]; b[s][1] = v[1];
                    } else {
                        b[s][0] = *wp;
                    }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][s], b[s][j], acc[mi][j], 0, 0, 0);
            }
        }
    }
    if (!active) return;

    // ---- epilogue: + bias (+ residual), vector store of NJ adjacent channels per row ----
    const int co = nb * NB + NJ * li;
    float bj[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bj[j] = p.bias[co + j];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave_row0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            const int t = t0 + row;
            if (t < p.L) {
                const size_t off = (seq_base + t) * p.cout_total + co;
                float v[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) v[j] = acc[mi][j][r] + bj[j];
                if (p.res) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) v[j] += p.res[off + j];
                }
                if constexpr (NJ == 4) {
                    *reinterpret_cast<f32x4*>(p.y + off) = f32x4{v[0], v[1], v[2], v[3]};
                } else if constexpr (NJ == 2) {
                    *reinterpret_cast<f32x2*>(p.y + off) = f32x2{v[0], v[1]};
                } else {
                    p.y[off] = v[0];
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution with fp32 operands split into bf16 hi + lo ("bf16x3"):
//     x*w ~= x_hi*w_hi + x_hi*w_lo + x_lo*w_hi,   x_hi = bf16(x), x_lo = bf16(x - x_hi)
// three v_mfma_f32_32x32x16_bf16 per 16-channel K slab, fp32 accumulate.  Each operand carries a
// 16-bit significand (relative product error ~2^-16; measured end-to-end error vs the fp32 oracle is
// ~1e-5 of max|y|, two orders inside the 1e-3 parity bar) at 16/3 = 5.3x the fp32-MFMA rate.
//
//   workgroup = 4 waves as WM (time) x WN (32-channel blocks); a wave owns MI 32x32 tiles stacked in time:
//   TM = WM*MI*32 rows, TN = WN*32 channels.  grid = (sequences * time tiles, ceil(n_blocks32/WN), branches).
//   LDS row = [hi: CH bf16 | lo: CH bf16 | 16 B pad]  (pitch/16 odd => ds_read_b128 conflict-free over rows);
//   LeakyReLU / MRF mean / zero padding / the hi-lo split all happen while staging.
//   A fragment (32x32x16): lane l -> row l&31, k = 8*(l>>5)+j  == 16 contiguous bytes of the staged row.
//   B fragments are pre-packed on the host in exactly the lane order the MFMA wants and streamed from
//   L2 with one global_load_dwordx4 per lane, prefetched one whole tap (NC16 K-slabs) ahead through a
//   register ring; the stream is contiguous across taps and chunks, so the prefetch runs through the
//   chunk barrier.  Waves that share time rows (same wm) re-read the same A rows from LDS; waves that
//   share channels (same wn) hit the same weight lines in L1/L2.
// ------------------------------------------------------------------------------------------------
template <int MI, int WM, int WN, int NC16>
__global__ __launch_bounds__(256) void conv_mfma_bf16x3_kernel(const MultiConvParams mp) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int TM = WM * MI * 32;
    constexpr int CH = NC16 * 16;          // channels per staged chunk
    constexpr int PITCH = CH * 4 + 16;     // bytes per LDS row
    constexpr int C4N = CH / 4;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];

    const ConvParams& p = mp.p[blockIdx.z];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int li = lane & 31;
    const int g = lane >> 5;

    const int seq = blockIdx.x / p.tiles_per_seq;
    const int t0 = (blockIdx.x % p.tiles_per_seq) * TM;
    const int nb = blockIdx.y * WN + wn;
    const bool active = nb < p.n_blocks32;
    const int phase = active ? nb / p.nb32_per_phase : 0;
    const int R = TM + p.halo;
    const size_t seq_base = (size_t)seq * p.L;
    const float slope = p.slope;

    f32x16 acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;

    // B stream of this wave's 32-channel block: [chunk][tap][c16][hi|lo] fragments of 64 x 16 B
    const bf16x8* wp = p.w16 + (size_t)(active ? nb : 0) * p.ntaps * (p.cin / 16) * 128 + lane;
    bf16x8 bq[NC16][2];
#pragma unroll
    for (int u = 0; u < NC16; ++u) {
        bq[u][0] = wp[u * 128];
        bq[u][1] = wp[u * 128 + 64];
    }
    wp += NC16 * 128;
    const int wave_row0 = wm * (MI * 32);
    const int roff0 = __builtin_amdgcn_readfirstlane(p.tap_off0[phase] - p.off_min);

    for (int c0 = 0; c0 < p.cin; c0 += CH) {
        __syncthreads();
        // ---- stage split(act(X))[rows, c0 : c0+CH]; loads are issued UB at a time before any is consumed ----
        constexpr int UB = 8;
        const int n_units = R * C4N;
        auto split_store = [&](int idx, const f32x4& v) {
            const int r = idx / C4N;
            const int c4 = idx - r * C4N;
            bf16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = lrelu(v[e], slope);
                hi[e] = (__bf16)a;
                lo[e] = (__bf16)(a - (float)hi[e]);
            }
            *reinterpret_cast<bf16x4*>(smem_b + r * PITCH + c4 * 8) = hi;
            *reinterpret_cast<bf16x4*>(smem_b + r * PITCH + CH * 2 + c4 * 8) = lo;
        };
        if (p.nin == 1) {
            for (int base = 0; base < n_units; base += 256 * UB) {
                f32x4 v[UB];
#pragma unroll
                for (int q = 0; q < UB; ++q) {
                    const int idx = base + q * 256 + tid;
                    const int r = idx / C4N;
                    const int c4 = idx - r * C4N;
                    const int t = t0 + p.off_min + r;
                    v[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (idx < n_units && t >= 0 && t < p.L)
                        v[q] = *reinterpret_cast<const f32x4*>(p.x0 + (seq_base + t) * p.cin + c0 + c4 * 4);
                }
#pragma unroll
                for (int q = 0; q < UB; ++q) {
                    const int idx = base + q * 256 + tid;
                    if (idx < n_units) split_store(idx, v[q]);
                }
            }
        } else {  // MRF mean of the previous stage's ResBlock outputs (upsample convs only)
            for (int idx = tid; idx < n_units; idx += 256) {
                const int r = idx / C4N;
                const int c4 = idx - r * C4N;
                const int t = t0 + p.off_min + r;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (t >= 0 && t < p.L) {
                    const size_t off = (seq_base + t) * p.cin + c0 + c4 * 4;
                    v = *reinterpret_cast<const f32x4*>(p.x0 + off);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(p.x1 + off);
                    if (p.nin == 3) {
                        const f32x4 v2 = *reinterpret_cast<const f32x4*>(p.x2 + off);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ((v[e] + v1[e]) + v2[e]) / 3.0f;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (v[e] + v1[e]) / 2.0f;
                    }
                }
                split_store(idx, v);
            }
        }
        __syncthreads();
        if (!active) continue;
        for (int t = 0; t < p.ntaps; ++t) {
            const int roff = roff0 + t * p.tap_step;
            const char* arow = smem_b + (wave_row0 + li + roff) * PITCH + g * 16;
#pragma unroll
            for (int u = 0; u < NC16; ++u) {
                const bf16x8 bh = bq[u][0];
                const bf16x8 bl = bq[u][1];
                bq[u][0] = wp[u * 128];       // same K slab of the next tap (or of the next chunk's first tap)
                bq[u][1] = wp[u * 128 + 64];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(arow + mi * 32 * PITCH + u * 32);
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(arow + mi * 32 * PITCH + CH * 2 + u * 32);
                    // D^T = W * X^T: rows = channels, cols = time, so a lane ends up with 4 adjacent channels per register quad
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al, acc[mi], 0, 0, 0);
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah, acc[mi], 0, 0, 0);
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, acc[mi], 0, 0, 0);
                }
            }
            wp += NC16 * 128;
        }
    }
    if (!active) return;

    // epilogue: lane holds time column t = li and channels co = 8q + 4g + {0..3} in acc[mi][4q .. 4q+3]
    f32x4 bias4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bias4[q] = *reinterpret_cast<const f32x4*>(p.bias + nb * 32 + 8 * q + 4 * g);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int t = t0 + wave_row0 + mi * 32 + li;
        if (t < p.L) {
            const size_t off = (seq_base + t) * p.cout_total + nb * 32 + 4 * g;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[mi][4 * q + e] + bias4[q][e];
                if (p.res) {
                    const f32x4 rv = *reinterpret_cast<const f32x4*>(p.res + off + 8 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rv[e];
                }
                *reinterpret_cast<f32x4*>(p.y + off + 8 * q) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Front end: PastFCEncoder (reference pytorch_layers.py:438-460) + feature transpose + AR concat
// (reference hifigan.py:208-211).  One workgroup per utterance.
//   xin[b, t, 0:cf]        = c[b, :, t]                (features, (B, C, T) -> channels-last)
//   xin[b, t, cf:cf+ar_out] = MLP(prev[b, :])          (time-constant AR channels)
//   xin[b, t, rest]        = 0                         (pad to a multiple of 16 channels)
// MLP weights are stored transposed (in, out): a lane reads 4 adjacent outputs (16 B) per input row, 1 KB per
// wave-instruction; the four waves split the input dimension and reduce through LDS.  The MLP sits on the
// AR critical path (chunk n+1 cannot start before it), so it is built for latency: 8 loads in flight per lane.
// ------------------------------------------------------------------------------------------------
struct FrontParams {
    const float* c;       // features; element (b, ch, t) at c[b*c_bstride + ch*c_cstride + t]
    int64_t c_bstride;
    int64_t c_cstride;
    const float* prev;    // AR context; element (b, i) at prev[b*prev_bstride + i]; null => zeros
    int64_t prev_bstride;
    float* xin;           // (B, T, cin_pad)
    int T;
    int cf;               // feature channels
    int cin_pad;
    int use_ar;
    int ar_input, ar_hidden, ar_output;
    const float* wt[5];   // transposed Linear weights (in, out)
    const float* bs[5];
};

__global__ __launch_bounds__(256) void front_kernel(const FrontParams p) {
    __shared__ __attribute__((aligned(16))) float act[2][1024];
    __shared__ __attribute__((aligned(16))) float part[4][1024];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int ks = tid >> 6;  // wave index = K slice: each wave reduces a quarter of the input dimension
    int cur = 0;
    if (p.use_ar) {
        for (int i = tid; i < p.ar_input; i += 256) act[0][i] = p.prev ? p.prev[(size_t)b * p.prev_bstride + i] : 0.f;
        __syncthreads();
        int din = p.ar_input;
        for (int layer = 0; layer < 5; ++layer) {
            const int dout = layer == 4 ? p.ar_output : p.ar_hidden;
            const float* wt = p.wt[layer];
            const float* x = act[cur];
            const int i0 = (din * ks) / 4, i1 = (din * (ks + 1)) / 4;
            for (int j4 = lane * 4; j4 < dout; j4 += 256) {  // dims are multiples of 4 (checked at create)
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
                int i = i0;
                for (; i + 8 <= i1; i += 8) {  // 8 independent 16-byte loads in flight per lane
                    f32x4 w[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) w[q] = *reinterpret_cast<const f32x4*>(wt + (size_t)(i + q) * dout + j4);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float xv = x[i + q];
#pragma unroll
                        for (int e = 0; e < 4; ++e) s[e] = fmaf(xv, w[q][e], s[e]);
                    }
                }
                for (; i < i1; ++i) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wt + (size_t)i * dout + j4);
                    const float xv = x[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[e] = fmaf(xv, w[e], s[e]);
                }
                *reinterpret_cast<f32x4*>(&part[ks][j4]) = s;
            }
            __syncthreads();
            for (int j = tid; j < dout; j += 256) {
                const float v = ((part[0][j] + part[1][j]) + (part[2][j] + part[3][j])) + p.bs[layer][j];
                act[cur ^ 1][j] = layer < 4 ? lrelu(v, 0.1f) : v;
            }
            __syncthreads();
            cur ^= 1;
            din = dout;
        }
    }
    // act[cur][0:ar_output] now holds the AR features
    const float* feats = act[cur];
    const int n = p.T * p.cin_pad;
    float* xo = p.xin + (size_t)b * n;
    for (int idx = tid; idx < n; idx += 256) {
        const int t = idx / p.cin_pad;
        const int ch = idx - t * p.cin_pad;
        float v = 0.f;
        if (ch < p.cf) v = p.c[(size_t)b * p.c_bstride + (size_t)ch * p.c_cstride + t];
        else if (p.use_ar && ch < p.cf + p.ar_output) v = feats[ch - p.cf];
        xo[idx] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Output conv: LeakyReLU(0.01) -> Conv1d(C -> 1, k) -> tanh (reference hifigan.py:146-159, 231), reading
// the MRF mean of the last stage's three ResBlock outputs on the fly.  C*k = 224 MACs per sample
// (0.015 % of the path): a plain VALU kernel, one output sample per thread, input rows staged in LDS.
// Writes the waveform in the boundary layout: out[b*out_bstride + t].
// ------------------------------------------------------------------------------------------------
struct OutConvParams {
    const float* x0;
    const float* x1;
    const float* x2;
    int nin;           // inputs averaged (1..3), as in ConvParams
    const float* w;    // [k][C]
    float bias;
    float* out;
    int64_t out_bstride;
    int L;             // samples per sequence
    int C;
    int K;
    float slope;
    int use_tanh;
};

__global__ __launch_bounds__(256) void output_conv_kernel(const OutConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int seq = blockIdx.y;
    const int t0 = blockIdx.x * 256;
    const int pad = (p.K - 1) / 2;
    const int R = 256 + p.K - 1;
    const int P = p.C + 1;
    float* ws = smem + R * P;  // weights [k][C]
    for (int i = tid; i < p.K * p.C; i += 256) ws[i] = p.w[i];
    const size_t base = (size_t)seq * p.L;
    for (int idx = tid; idx < R * p.C; idx += 256) {
        const int r = idx / p.C;
        const int ch = idx - r * p.C;
        const int t = t0 - pad + r;
        float v = 0.f;
        if (t >= 0 && t < p.L) {
            const size_t off = (base + t) * p.C + ch;
            v = p.x0[off];
            if (p.nin == 3) v = ((v + p.x1[off]) + p.x2[off]) / 3.0f;
            else if (p.nin == 2) v = (v + p.x1[off]) / 2.0f;
            v = lrelu(v, p.slope);
        }
        smem[r * P + ch] = v;
    }
    __syncthreads();
    const int t = t0 + tid;
    if (t < p.L) {
        float s = p.bias;
        for (int k = 0; k < p.K; ++k) {
            const float* xr = &smem[(tid + k) * P];
            const float* wk = &ws[k * p.C];
            for (int ch = 0; ch < p.C; ++ch) s = fmaf(xr[ch], wk[ch], s);
        }
        p.out[(size_t)seq * p.out_bstride + t] = p.use_tanh ? tanhf(s) : s;
    }
}

}  // namespace hificar
