// The generator's small kernels (one translation unit: hificar.hip): front_kernel (PastFCEncoder + input assembly), mrf_split_kernel,
// output_conv_kernel, tap / phoneme-head / PCM helpers.  The conv engine's templates are in hificar_conv.hip.h (instantiated per set in
// hificar_conv_inst.hip); layout conventions are described there.
#pragma once
#include "hificar_conv.hip.h"

namespace hificar {

// MRF mean + LeakyReLU + split for the upsample convs' input: out = split(lrelu((((x0 + x1) + x2) + x3) / n, slope)) over the n <= 4 blocks
// (hifigan.py:226-230: cs = 0.0; cs += block_j(c) in order; c = cs / n).
// Elementwise, HBM-bound; one thread = 8 channels (2 x 16 B in per input, 16 B hi + 16 B lo out).
struct MrfSplitParams {
    const float* x0;
    const float* x1;
    const float* x2;
    const float* x3;  // (a fourth residual block per stage: hifigan.py:134-145 builds one per resblock_kernel_sizes entry)
    char* out;
    int nin;
    int C;
    long long rows;
    float slope;
    int f32;  // 1: write plain fp32 rows of LeakyReLU(mean) (exact-fp32 arithmetic) instead of split rows
};

__global__ __launch_bounds__(256) void mrf_split_kernel(const MrfSplitParams p) {
    const int c8n = p.C >> 3;
    const long long total = p.rows * c8n;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long row = idx / c8n;
        const int c8 = (int)(idx - row * c8n);
        const size_t off = (size_t)row * p.C + c8 * 8;
        float v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 a = *reinterpret_cast<const f32x4*>(p.x0 + off + 4 * h);
            if (p.nin >= 2) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(p.x1 + off + 4 * h);
                if (p.nin == 4) {
                    const f32x4 c = *reinterpret_cast<const f32x4*>(p.x2 + off + 4 * h);
                    const f32x4 d = *reinterpret_cast<const f32x4*>(p.x3 + off + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = (((a[e] + b[e]) + c[e]) + d[e]) / 4.0f;
                } else if (p.nin == 3) {
                    const f32x4 c = *reinterpret_cast<const f32x4*>(p.x2 + off + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = ((a[e] + b[e]) + c[e]) / 3.0f;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = (a[e] + b[e]) / 2.0f;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * h + e] = a[e];
        }
        char* orow = p.out + (size_t)row * p.C * 4;
        if (p.f32) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], v[e] * p.slope);
            *reinterpret_cast<f32x4*>(orow + c8 * 32) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(orow + c8 * 32 + 16) = f32x4{v[4], v[5], v[6], v[7]};
            continue;
        }
        bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = fmaxf(v[e], v[e] * p.slope);
            hi[e] = (__bf16)a;
            lo[e] = (__bf16)(a - (float)hi[e]);
        }
        *reinterpret_cast<bf16x8*>(orow + c8 * 16) = hi;
        *reinterpret_cast<bf16x8*>(orow + p.C * 2 + c8 * 16) = lo;
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
    float* xin;           // (B, T, cin_pad) fp32 rows, or null
    char* xin_s;          // (B, T, [hi cin_pad | lo cin_pad]) split rows (no activation: the input conv has none), or null
    int T;
    int cf;               // feature channels
    int cin_pad;
    int use_ar;
    int ar_input, ar_hidden, ar_output;
    const float* wt[5];   // transposed Linear weights (in, out)
    const float* bs[5];
    // packed mode (hificar_ar_loop_packed): sequence b of this step is utterance slots[b].x continuing at frame slots[b].y
    // with valid[b] frames; c / prev are then the bases of the packed feature / waveform tensors (prev = waveform, row
    // pitch prev_bstride, hop samples per frame): its AR context are the ar_input samples before hop * frame.
    const int2* slots;
    const int* valid;
    int hop;
    int t_valid;          // frames that exist in c (<= T): later frames read as zeros (bucketed launch lengths)
    // speaker conditioning (hifigan.py:212-216): spk_fc(spk_emb_mat[spk_id[b]]) is added to channels [0, cf + ar_output)
    const int* spk_id;    // (B) or null
    const float* spk_emb; // (num_spk, spk_e)
    const float* spk_w;   // spk_fc.weight (cf + ar_output, spk_e)
    const float* spk_b;   // spk_fc.bias
    int spk_e;
    // phoneme conditioning (hifigan.py:217-220): channels [cf + ar_output, + ph_e) of frame t hold ph_emb[ph[b, t]]
    const int* ph;        // (B, ph_stride) or null
    int ph_stride;
    const float* ph_emb;  // (num_ph, ph_e)
    int ph_e;
    float* mlp_tape;      // training: (B, 5, 1024) inputs of the five Linear layers (backward needs them), or null
};

constexpr int kFrontThreads = 1024;
__global__ __launch_bounds__(kFrontThreads) void front_kernel(const FrontParams p) {
    constexpr int NW = kFrontThreads / 64;  // waves = K slices: 16 slices x 16 loads in flight per lane cover a 512-input layer in two rounds
    constexpr int NT = kFrontThreads;
    __shared__ __attribute__((aligned(16))) float act[2][1024];
    __shared__ __attribute__((aligned(16))) float part[NW][512];  // layer outputs: ar_hidden, ar_output <= 512 (hificar_create)
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int ks = tid >> 6;  // wave index = K slice: each wave reduces an eighth of the input dimension
    int cur = 0;
    if (p.use_ar) {
        const float* prevp = p.prev ? p.prev + (size_t)b * p.prev_bstride : nullptr;
        if (p.slots) {
            const int2 sl = p.slots[b];
            prevp = sl.y > 0 ? p.prev + (size_t)sl.x * p.prev_bstride + (size_t)p.hop * sl.y - p.ar_input : nullptr;
        }
        for (int i = tid; i < p.ar_input; i += NT) {
            const float v = prevp ? prevp[i] : 0.f;
            act[0][i] = v;
            if (p.mlp_tape) p.mlp_tape[((size_t)b * 5) * 1024 + i] = v;
        }
        __syncthreads();
        int din = p.ar_input;
        for (int layer = 0; layer < 5; ++layer) {
            const int dout = layer == 4 ? p.ar_output : p.ar_hidden;
            const float* wt = p.wt[layer];
            const float* x = act[cur];
            const int i0 = (din * ks) / NW, i1 = (din * (ks + 1)) / NW;
            for (int j4 = lane * 4; j4 < dout; j4 += 256) {  // dims are multiples of 4 (checked at create)
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
                int i = i0;
                for (; i + 16 <= i1; i += 16) {  // 16 independent 16-byte loads in flight per lane
                    f32x4 w[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) w[q] = *reinterpret_cast<const f32x4*>(wt + (size_t)(i + q) * dout + j4);
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
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
            for (int j = tid; j < dout; j += NT) {
                float v = p.bs[layer][j];
#pragma unroll
                for (int q = 0; q < NW; ++q) v += part[q][j];
                const float a = layer < 4 ? lrelu(v, 0.1f) : v;
                act[cur ^ 1][j] = a;
                if (p.mlp_tape && layer < 4) p.mlp_tape[((size_t)b * 5 + layer + 1) * 1024 + j] = a;
            }
            __syncthreads();
            cur ^= 1;
            din = dout;
        }
    }
    // act[cur][0:ar_output] now holds the AR features
    const float* feats = act[cur];
    const int c_ar = p.use_ar ? p.ar_output : 0;
    float* spk_vec = act[cur ^ 1];  // free once the MLP is done (in_channels <= 1024 with use_spk_id, hificar_create)
    if (p.spk_id) {  // per-utterance speaker vector for channels [0, cf + ar_output)
        const float* e = p.spk_emb + (size_t)p.spk_id[b] * p.spk_e;
        for (int ch = tid; ch < p.cf + c_ar; ch += NT) {
            float v = p.spk_b[ch];
            for (int k = 0; k < p.spk_e; ++k) v = fmaf(p.spk_w[(size_t)ch * p.spk_e + k], e[k], v);
            spk_vec[ch] = v;
        }
        __syncthreads();
    }
    const int n = p.T * p.cin_pad;
    size_t cbase = (size_t)b * p.c_bstride;
    int tmax = p.t_valid;
    if (p.slots) {
        cbase = (size_t)p.slots[b].x * p.c_bstride + p.slots[b].y;
        tmax = p.valid[b];  // frames past the utterance's end may lie outside the packed tensor
    }
    for (int idx = tid; idx < n; idx += NT) {
        const int t = idx / p.cin_pad;
        const int ch = idx - t * p.cin_pad;
        float v = 0.f;
        if (ch < p.cf) v = t < tmax ? p.c[cbase + (size_t)ch * p.c_cstride + t] : 0.f;
        else if (ch < p.cf + c_ar) v = feats[ch - p.cf];
        else if (p.ph && ch < p.cf + c_ar + p.ph_e) v = t < tmax ? p.ph_emb[(size_t)p.ph[(size_t)b * p.ph_stride + t] * p.ph_e + (ch - p.cf - c_ar)] : 0.f;
        if (p.spk_id && ch < p.cf + c_ar && t < tmax) v += spk_vec[ch];
        if (p.xin) p.xin[(size_t)b * n + idx] = v;
        if (p.xin_s) {
            __bf16* row = reinterpret_cast<__bf16*>(p.xin_s + ((size_t)b * p.T + t) * p.cin_pad * 4);
            const __bf16 hi = (__bf16)v;
            row[ch] = hi;
            row[p.cin_pad + ch] = (__bf16)(v - (float)hi);
        }
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
    const float* x3;
    int nin;           // inputs averaged (1..4), as in MrfSplitParams
    const float* w;    // [k][C]
    float bias;
    const float* bias_ptr;  // training: the bias lives on the device (overrides `bias`)
    float* out;
    int64_t out_bstride;
    int L;             // samples per sequence
    int C;
    int K;
    float slope;
    int use_tanh;
    const int* seq_len;  // ragged batches, as in ConvParams (rows = samples)
    int len_const;
    int len_f0, len_max, len_mul;
    const int2* slots;   // packed mode, as in FrontParams: sequence -> (utterance, first frame); out is the packed waveform
    int hop;
    int TR;              // output samples per workgroup (<= 256; fewer when C is wide, so that the LDS tile fits)
};

__global__ __launch_bounds__(256) void output_conv_kernel(const OutConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int seq = blockIdx.y;
    const int t0 = blockIdx.x * p.TR;
    const int pad = (p.K - 1) / 2;
    const int R = p.TR + p.K - 1;
    const int P = p.C + 1;
    float* ws = smem + R * P;  // weights [k][C]
    for (int i = tid; i < p.K * p.C; i += 256) ws[i] = p.w[i];
    const size_t base = (size_t)seq * p.L;
    const int Ls = (p.seq_len || p.len_const >= 0) ? min(max((p.seq_len ? p.seq_len[seq] : p.len_const) - p.len_f0, 0), p.len_max) * p.len_mul : p.L;
    if (t0 >= Ls) return;  // nothing of this sequence in the block (uniform)
    const int c4n = p.C >> 2;  // C is a multiple of 32: 16-byte loads
    for (int idx = tid; idx < R * c4n; idx += 256) {
        const int r = idx / c4n;
        const int ch = (idx - r * c4n) * 4;
        const int t = t0 - pad + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < Ls) {
            const size_t off = (base + t) * p.C + ch;
            v = *reinterpret_cast<const f32x4*>(p.x0 + off);
            if (p.nin >= 2) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(p.x1 + off);
                if (p.nin == 4) {
                    const f32x4 c = *reinterpret_cast<const f32x4*>(p.x2 + off);
                    const f32x4 d = *reinterpret_cast<const f32x4*>(p.x3 + off);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (((v[e] + b[e]) + c[e]) + d[e]) / 4.0f;
                } else if (p.nin == 3) {
                    const f32x4 c = *reinterpret_cast<const f32x4*>(p.x2 + off);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ((v[e] + b[e]) + c[e]) / 3.0f;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (v[e] + b[e]) / 2.0f;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = lrelu(v[e], p.slope);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) smem[r * P + ch + e] = v[e];
    }
    __syncthreads();
    const int t = t0 + tid;
    if (tid < p.TR && t < Ls) {
        float s = p.bias_ptr ? *p.bias_ptr : p.bias;
        for (int k = 0; k < p.K; ++k) {
            const float* xr = &smem[(tid + k) * P];
            const float* wk = &ws[k * p.C];
            for (int ch = 0; ch < p.C; ++ch) s = fmaf(xr[ch], wk[ch], s);
        }
        const size_t obase = p.slots ? (size_t)p.slots[seq].x * p.out_bstride + (size_t)p.hop * p.slots[seq].y : (size_t)seq * p.out_bstride;
        p.out[obase + t] = p.use_tanh ? tanhf(s) : s;
    }
}

// Debug tap (hificar_debug_tap): channels-last rows (pitch floats per row; channels c0 .. c0+C) -> (sequence, channel, row), the
// reference's (B, C, L) layout.  split = 1: the source holds split rows [hi | lo] of bf16 (|pitch| channels each): value = hi + lo.
struct TapParams {
    const void* src;
    float* dst;
    int pitch;  // floats per source row (negative: split rows of -pitch channels)
    int c0, C, rows;
    int src_rows;     // rows per sequence in the source (>= rows)
    long long total;  // sequences * C * rows
    int split;
};

__global__ __launch_bounds__(256) void tap_copy_kernel(const TapParams p) {
    const int pitch = p.pitch < 0 ? -p.pitch : p.pitch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i % p.rows);
        const long long bc = i / p.rows;
        const int c = (int)(bc % p.C);
        const long long b = bc / p.C;
        const size_t row = (size_t)b * p.src_rows + r;
        float v;
        if (p.split) {
            const __bf16* rp = reinterpret_cast<const __bf16*>(p.src) + row * pitch * 2;
            v = (float)rp[p.c0 + c] + (float)rp[pitch + p.c0 + c];
        } else {
            v = reinterpret_cast<const float*>(p.src)[row * pitch + p.c0 + c];
        }
        p.dst[i] = v;
    }
}

// Phoneme-loss head (hifigan.py:183-189, 232-237): ph_out[b, p, f] = AvgPool1d(kernel 2*hop, stride hop, padding hop/2)(ph_fc(c))
// with c the last stage's MRF mean.  ph_fc is linear, so the window's mean of c goes through it once: one workgroup per
// (frame, utterance) sums the window's rows (zero padding counts in the divisor, as AvgPool1d's count_include_pad default does; the
// bias only where a sample exists) and applies the (num_ph x C) matrix.
struct PhHeadParams {
    const float* x0;
    const float* x1;
    const float* x2;
    const float* x3;
    int nin;          // ResBlock outputs averaged (1..4)
    const float* w;   // ph_fc.weight (num_ph, C)
    const float* bias;
    float* out;       // (B, num_ph, T)
    int C;            // real channels
    int Cp;           // row pitch (padded channels)
    int L;            // rows (samples) per sequence in the stage buffers
    int T;            // output frames per utterance (row pitch of out)
    int hop;
    int num_ph;
    const int* seq_len;
    int len_const;
};

__global__ __launch_bounds__(256) void ph_head_kernel(const PhHeadParams p) {
    __shared__ float part[8][128];
    __shared__ float mean[128];
    const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int frames = p.seq_len ? p.seq_len[b] : (p.len_const >= 0 ? p.len_const : p.T);
    if (f >= frames) return;  // frames past the utterance's end are not written
    const int Ls = frames * p.hop;
    const int t0 = f * p.hop - p.hop / 2, K = 2 * p.hop;
    const int lo = max(t0, 0), hi = min(t0 + K, Ls);
    const int ch = tid & 31, rs = tid >> 5;  // 8 row slices x 32 channels per pass (C <= 128: up to 4 channel passes)
    for (int c0 = 0; c0 < p.C; c0 += 32) {
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
    __syncthreads();
    for (int c = tid; c < p.C; c += 256) {
        float s = 0.f;
        for (int r = 0; r < 8; ++r) s += part[r][c];
        mean[c] = s;
    }
    __syncthreads();
    for (int q = tid; q < p.num_ph; q += 256) {
        float s = p.bias[q] * (float)(hi - lo);
        for (int c = 0; c < p.C; ++c) s = fmaf(p.w[(size_t)q * p.C + c], mean[c], s);
        p.out[((size_t)b * p.num_ph + q) * p.T + f] = s / (float)K;
    }
}

// float waveform -> PCM_16 (reference: sf.write(..., "PCM_16") on the host, decode.py:319-324)
__global__ __launch_bounds__(256) void pcm16_kernel(const float* __restrict__ x, int16_t* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        // libsndfile's float32 -> PCM_16 (src/pcm.c f2s_array): lrintf(src * 32767.f) — a float32 product, round to nearest even;
        // clipped here where libsndfile (clipping off) would wrap.  Bit-identical to the host writer (bin/predict_wav.py::write_wav).
        const float v = rintf(x[i] * 32767.0f);
        y[i] = (int16_t)fminf(fmaxf(v, -32768.0f), 32767.0f);
    }
}

}  // namespace hificar
