// Discriminator kernels of libhificar (HiFiGANMultiScaleMultiPeriodDiscriminator, reference articulatory/models/hifigan.py:317-825):
// every Conv1d / Conv2d (k, 1) of the discriminators runs as ONE GEMM per group on the generator's exact-fp32 conv kernels
// (conv_f32_kernel with a single tap: A [M][Kg] x W [Kg][N]) over an im2col matrix whose rows are ALL output positions of ALL sequences
// — the period discriminators' columns are 3..1260-row sequences (B x period of them), far too short to tile one by one.  This file holds
// the gathers around those GEMMs; all of them are HBM-bound element-wise passes.
//   im2col_kernel        A[m = (seq, t_out)][j = tap * cin_g + c] = in[seq][t_out * stride + tap - pad][ci0 + c]   (0 outside)
//                        `in` = the raw signal (scale form, or period form with the reference's reflect padding) or the previous
//                        layer's activated output (one [M][Np] buffer per group)
//   col2im_mask_kernel   backward of im2col fused with LeakyReLU' and the loss's own gradient of that feature map:
//                        dZ_prev[m][n] = (dY_prev[m][n] + sum_{taps} dA[(seq, t_out)][tap * cin_g + c]) * lrelu'(Y_prev[m][n])
//   signal_grad_kernel   the same gather for layer 0: gradient of the raw signal (reflect padding folded back), accumulated over the
//                        sub-discriminators in launch order (deterministic)
//   avgpool_kernel / avgpool_bwd_kernel   AvgPool1d(kernel, stride, padding), count_include_pad (hifigan.py:700-738)
#pragma once

namespace hificar {

struct Im2colParams {
    const float* src;
    float* dst;       // group g: dst + g * dst_gstride, [M][Kg_pad]
    long long dst_gstride;
    long long per_group4;  // M * Kg_pad / 4
    long long total4; // groups * M * Kg_pad / 4
    int Kg, Kg_pad;
    int L_in, L_out;  // rows per sequence
    int kt, stride, pad, cin_g;
    int src_mode;     // 0 raw signal, scale form: seq = b;  1 raw signal, period form: seq = (b, col);  2 previous layer
    int T, period;    // raw signal length (unpadded); period
    int ci0;          // (device side: first input channel of the group being gathered = group * cin_g)
    int prev_cout_g, prev_np;
    long long prev_gstride;  // floats between the previous layer's group buffers
};

__device__ __forceinline__ float im2col_fetch(const Im2colParams& p, int seq, int t, int c) {
    if (t < 0 || t >= p.L_in) return 0.f;
    if (p.src_mode == 0) return p.src[(size_t)seq * p.T + t];
    if (p.src_mode == 1) {
        const int b = seq / p.period, col = seq - b * p.period;
        int idx = t * p.period + col;
        if (idx >= p.T) idx = 2 * (p.T - 1) - idx;  // F.pad(..., "reflect") on the right (hifigan.py:404-407)
        return p.src[(size_t)b * p.T + idx];
    }
    const int ca = p.ci0 + c;
    const int g = ca / p.prev_cout_g, n = ca - g * p.prev_cout_g;
    return p.src[(size_t)g * p.prev_gstride + ((size_t)seq * p.L_in + t) * p.prev_np + n];
}

__global__ __launch_bounds__(256) void im2col_kernel(const Im2colParams q) {
    Im2colParams p = q;
    const int k4 = p.Kg_pad >> 2;
    const bool vec = p.src_mode == 2 && (p.cin_g & 3) == 0 && (p.prev_cout_g & 3) == 0;
    for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < p.total4; i0 += (long long)gridDim.x * 256) {
        const int grp = (int)(i0 / p.per_group4);
        const long long i = i0 - (long long)grp * p.per_group4;
        p.ci0 = grp * p.cin_g;
        const int m = (int)(i / k4), j = (int)(i - (long long)m * k4) * 4;
        const int seq = m / p.L_out, to = m - seq * p.L_out;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (j < p.Kg) {
            if (vec) {
                const int tap = j / p.cin_g, c = j - tap * p.cin_g;
                const int t = to * p.stride + tap - p.pad;
                if (t >= 0 && t < p.L_in) {
                    const int ca = p.ci0 + c;
                    const int g = ca / p.prev_cout_g, n = ca - g * p.prev_cout_g;
                    v = *reinterpret_cast<const f32x4*>(p.src + (size_t)g * p.prev_gstride + ((size_t)seq * p.L_in + t) * p.prev_np + n);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int jj = j + u;
                    if (jj < p.Kg) {
                        const int tap = jj / p.cin_g, c = jj - tap * p.cin_g;
                        v[u] = im2col_fetch(p, seq, to * p.stride + tap - p.pad, c);
                    }
                }
            }
        }
        reinterpret_cast<f32x4*>(p.dst + (size_t)grp * p.dst_gstride)[i] = v;
    }
}

// The previous-layer gather of im2col_kernel (src_mode 2, 4 | cin_g and 4 | prev_cout_g) with blockIdx.y = group and 32-bit index arithmetic
// inside a group: no 64-bit divisions, the group decode is free.  Pure copies: the same matrix.
__global__ __launch_bounds__(256) void im2col4_kernel(const Im2colParams p) {
    const int grp = blockIdx.y;
    const int k4 = p.Kg_pad >> 2;
    const int n4 = (int)p.per_group4;
    const int ci0 = grp * p.cin_g;
    f32x4* const dst = reinterpret_cast<f32x4*>(p.dst + (size_t)grp * p.dst_gstride);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const int m = i / k4, j = (i - m * k4) * 4;
        const int seq = m / p.L_out, to = m - seq * p.L_out;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (j < p.Kg) {
            const int tap = j / p.cin_g, c = j - tap * p.cin_g;
            const int t = to * p.stride + tap - p.pad;
            if (t >= 0 && t < p.L_in) {
                const int ca = ci0 + c;
                const int g = ca / p.prev_cout_g, n = ca - g * p.prev_cout_g;
                v = *reinterpret_cast<const f32x4*>(p.src + (size_t)g * p.prev_gstride + ((size_t)seq * p.L_in + t) * p.prev_np + n);
            }
        }
        dst[i] = v;
    }
}

// dZ of the PREVIOUS layer (one launch per previous-layer group buffer) from this layer's dA buffers.
struct Col2imParams {
    const float* dA;        // this layer's dA buffers: group g at dA + g * da_gstride, [M][Kg_pad]
    long long da_gstride;
    const float* dy[16];    // per previous-layer group: the loss's gradient of that output buffer [Mp][Np] or nullptr
    const float* y;         // the previous layer's activated outputs: group gp at y + gp * y_gstride, [Mp][Np]
    float* dz;              // out, same strides
    long long y_gstride;
    long long per_group;    // Mp * Np
    long long total;        // groups_prev * Mp * Np
    int np, cout_g_prev;    // previous layer's pitch / channels per group
    int L_in, L_out;        // rows per sequence of the previous layer's output (= this layer's input) / of this layer's output
    int kt, stride, pad, cin_g, Kg_pad;
    float slope;
    // sliding-window layers: dA holds the data gradient itself, dX [group][seq][win_rows][win_pitch] (the transposed conv's output), not
    // the gradient of an im2col matrix: one read instead of the tap sum
    int win, win_rows, win_pitch, win_off;  // (win_off: dX rows are PADDED positions, t + win_off)
};

__global__ __launch_bounds__(256) void col2im_mask_kernel(const Col2imParams p) {
    for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < p.total; i0 += (long long)gridDim.x * 256) {
        const int gp = (int)(i0 / p.per_group);
        const long long i = i0 - (long long)gp * p.per_group;
        const int mp = (int)(i / p.np), n = (int)(i - (long long)mp * p.np);
        const float* dyg = p.dy[gp];
        float r = 0.f;
        if (n < p.cout_g_prev) {
            const int seq = mp / p.L_in, t = mp - seq * p.L_in;
            const int ca = gp * p.cout_g_prev + n;
            const int g = ca / p.cin_g, c = ca - g * p.cin_g;
            const float* da = p.dA + (size_t)g * p.da_gstride;
            float s = dyg ? dyg[i] : 0.f;
            if (p.win) s += da[((size_t)seq * p.win_rows + t + p.win_off) * p.win_pitch + c];
            // taps with (t + pad - tap) divisible by the stride and the output position in range, ascending tap order
            for (int tap = p.win ? p.kt : (t + p.pad) % p.stride; tap < p.kt; tap += p.stride) {
                const int num = t + p.pad - tap;
                if (num < 0) break;
                const int to = num / p.stride;
                if (to < p.L_out) s += da[((size_t)seq * p.L_out + to) * p.Kg_pad + tap * p.cin_g + c];
            }
            const float yv = p.y[(size_t)gp * p.y_gstride + i];
            r = yv > 0.f ? s : s * p.slope;  // LeakyReLU'(x <= 0) = slope; y = 0 only where x = 0
        }
        p.dz[(size_t)gp * p.y_gstride + i] = r;
    }
}

// The same on four adjacent channels per thread (16-byte loads and stores, one row / group decode per four elements, 32-bit index
// arithmetic inside a group): blockIdx.y = previous-layer group.  Needs 4 | np, cout_g_prev, cin_g, Kg_pad, every stride and 16-byte aligned
// bases (checked by the launcher, which otherwise takes the scalar kernel).  Each element's sum runs in the scalar kernel's order (loss
// gradient, window read, taps ascending): bit-identical results.
__global__ __launch_bounds__(256) void col2im_mask4_kernel(const Col2imParams p) {
    const int gp = blockIdx.y;
    const int nvec = (int)(p.per_group >> 2);
    const float* dyg = p.dy[gp];
    const float* yg = p.y + (size_t)gp * p.y_gstride;
    float* dzg = p.dz + (size_t)gp * p.y_gstride;
    const int np4 = p.np >> 2;
    for (int iv = blockIdx.x * 256 + threadIdx.x; iv < nvec; iv += gridDim.x * 256) {
        const int mp = iv / np4, n = (iv - mp * np4) << 2;
        const size_t i = (size_t)iv << 2;
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (n < p.cout_g_prev) {
            const int seq = mp / p.L_in, t = mp - seq * p.L_in;
            const int ca = gp * p.cout_g_prev + n;
            const int g = ca / p.cin_g, c = ca - g * p.cin_g;
            const float* da = p.dA + (size_t)g * p.da_gstride;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            if (dyg) s = *reinterpret_cast<const f32x4*>(dyg + i);
            if (p.win) s += *reinterpret_cast<const f32x4*>(da + ((size_t)seq * p.win_rows + t + p.win_off) * p.win_pitch + c);
            for (int tap = p.win ? p.kt : (t + p.pad) % p.stride; tap < p.kt; tap += p.stride) {
                const int num = t + p.pad - tap;
                if (num < 0) break;
                const int to = num / p.stride;
                if (to < p.L_out) s += *reinterpret_cast<const f32x4*>(da + ((size_t)seq * p.L_out + to) * p.Kg_pad + tap * p.cin_g + c);
            }
            const f32x4 yv = *reinterpret_cast<const f32x4*>(yg + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = yv[e] > 0.f ? s[e] : s[e] * p.slope;
        }
        *reinterpret_cast<f32x4*>(dzg + i) = r;
    }
}

// Gradient of the raw signal from a first layer's dA ([M][Kg_pad], cin = 1): dx[b][i] (+)= sum over the positions that read sample i.
struct SignalGradParams {
    const float* dA;
    float* dx;        // (B, T_sub) (scale form: the pooled signal of that scale; period form: the raw signal)
    long long total;  // B * T
    int T, period;    // period 0: scale form
    int L_in, L_out, kt, stride, pad, Kg_pad;
    int accumulate;   // 0: dx = ..., 1: dx += ...
};

__device__ __forceinline__ float signal_grad_at(const SignalGradParams& p, int seq, int t) {
    float s = 0.f;
    for (int tap = (t + p.pad) % p.stride; tap < p.kt; tap += p.stride) {
        const int num = t + p.pad - tap;
        if (num < 0) break;
        const int to = num / p.stride;
        if (to < p.L_out) s += p.dA[((size_t)seq * p.L_out + to) * p.Kg_pad + tap];
    }
    return s;
}

__global__ __launch_bounds__(256) void signal_grad_kernel(const SignalGradParams p) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / p.T), idx = (int)(i - (long long)b * p.T);
        float s;
        if (p.period == 0) {
            s = signal_grad_at(p, b, idx);
        } else {
            s = signal_grad_at(p, b * p.period + idx % p.period, idx / p.period);
            const int mir = 2 * (p.T - 1) - idx;  // the reflect-padded position that mirrors this sample, if any
            if (mir >= p.T && mir < p.L_in * p.period) s += signal_grad_at(p, b * p.period + mir % p.period, mir / p.period);
        }
        p.dx[i] = p.accumulate ? p.dx[i] + s : s;
    }
}

struct PoolParams {
    const float* src;
    float* dst;
    long long total;  // B * L_dst
    int L_src, L_dst, kernel, stride, pad;
    int accumulate;
};

__global__ __launch_bounds__(256) void avgpool_kernel(const PoolParams p) {  // dst = AvgPool1d(src), zeros count (count_include_pad)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / p.L_dst), o = (int)(i - (long long)b * p.L_dst);
        float s = 0.f;
        for (int k = 0; k < p.kernel; ++k) {
            const int t = o * p.stride + k - p.pad;
            if (t >= 0 && t < p.L_src) s += p.src[(size_t)b * p.L_src + t];
        }
        p.dst[i] = s / (float)p.kernel;
    }
}

// dst (B, L_dst = the pool's INPUT length) (+)= the pool's backward of src (B, L_src = its output length)
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const PoolParams p) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / p.L_dst), t = (int)(i - (long long)b * p.L_dst);
        float s = 0.f;
        for (int k = (t + p.pad) % p.stride; k < p.kernel; k += p.stride) {
            const int num = t + p.pad - k;
            if (num < 0) break;
            const int o = num / p.stride;
            if (o < p.L_src) s += p.src[(size_t)b * p.L_src + o];
        }
        s /= (float)p.kernel;
        p.dst[i] = p.accumulate ? p.dst[i] + s : s;
    }
}

// dst = src[0] + src[1] + ... in that order (n >= 0; n = 0: zeros)
struct SumParams {
    const float* src[9];  // HIFICAR_DISC_MAX_SUBS + 1
    float* dst;
    long long total;
    int n;
};

__global__ __launch_bounds__(256) void sum_kernel(const SumParams p) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < p.n; ++k) s += p.src[k][i];
        p.dst[i] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// GAN losses on the engine's own output buffers (articulatory/losses/adversarial_loss.py:12-123, feat_match_loss.py:12-54): every term
// is a mean over the elements of one layer output, so one pass per output buffer gives both the term's sum and its gradient.
//   adv_kind  0 none | 1 (a - 1)^2 | 2 a^2 | 3 -a | 4 -min(a - 1, 0) | 5 -min(-a - 1, 0)      (mse real / gen, mse fake, hinge gen, real, fake)
//   c_fm != 0: |a - b| against the reference pass's buffer at the same offset (feature matching; b is a constant)
// dout = w_adv * d(adv term) + w_fm * sign(a - b); partial sums per (entry, chunk) are combined in a fixed order by loss_reduce_kernel.
// ------------------------------------------------------------------------------------------------
constexpr int kLossChunks = 32;

struct LossEntry {
    long long y_off;   // floats from the tape base (both passes share the layout)
    long long d_off;   // floats from the gradient buffer's base
    long long total;   // rows * pitch
    int pitch, channels;
    int adv_kind;
    float w_adv, w_fm;   // d(total loss) / d(element) scales (lambda, averaging and 1 / numel folded in)
    float c_adv, c_fm;   // value scales (averaging and 1 / numel; no lambda)
};

__global__ __launch_bounds__(256) void loss_kernel(const LossEntry* entries, const float* tape, const float* tape_ref, float* douts, float* partials) {
    __shared__ float red[2][4];
    const LossEntry e = entries[blockIdx.y];
    const float* a = tape + e.y_off;
    const float* b = tape_ref ? tape_ref + e.y_off : nullptr;
    float* d = douts + e.d_off;
    const long long n4 = e.total >> 2;
    const long long lo = n4 * blockIdx.x / kLossChunks, hi = n4 * (blockIdx.x + 1) / kLossChunks;
    float s_adv = 0.f, s_fm = 0.f;
    const bool use_b = b && e.c_fm != 0.f;
    // one element quad: the same arithmetic, in the same order per thread, as ever (the partial sums — and with them every logged loss — do not move)
    auto quad = [&](long long i, int col, const f32x4& av, const f32x4& bv) {
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (col + u >= e.channels) continue;
            const float x = av[u];
            float ga = 0.f;
            switch (e.adv_kind) {
                case 1: s_adv += (x - 1.f) * (x - 1.f); ga = 2.f * (x - 1.f); break;
                case 2: s_adv += x * x; ga = 2.f * x; break;
                case 3: s_adv -= x; ga = -1.f; break;
                case 4: s_adv -= fminf(x - 1.f, 0.f); ga = x < 1.f ? -1.f : 0.f; break;
                case 5: s_adv -= fminf(-x - 1.f, 0.f); ga = x > -1.f ? 1.f : 0.f; break;
                default: break;
            }
            float gf = 0.f;
            if (e.c_fm != 0.f) {  // (feature matching takes part; w_fm may be 0 when a lambda is 0: the value is still logged)
                const float df = x - bv[u];
                s_fm += fabsf(df);
                gf = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
            }
            g[u] = e.w_adv * ga + e.w_fm * gf;
        }
        reinterpret_cast<f32x4*>(d)[i] = g;
    };
    if ((e.pitch & 3) == 0) {
        // Round 6: the column of a quad is carried along instead of a 64-bit `%` per quad (the loop was bound by that division: 2 TB/s on a pure
        // streaming pass), and four quads are in flight per thread.
        const int p4 = e.pitch >> 2, step4 = 256 % p4;
        int col4 = (int)((lo + threadIdx.x) % p4);
        auto advance = [&](int c) {
            c += step4;
            return c >= p4 ? c - p4 : c;
        };
        constexpr int UQ = 4;
        long long i = lo + threadIdx.x;
        for (; i + (UQ - 1) * 256 < hi; i += UQ * 256) {
            f32x4 av[UQ], bv[UQ];
            int cols[UQ];
#pragma unroll
            for (int q = 0; q < UQ; ++q) {
                av[q] = reinterpret_cast<const f32x4*>(a)[i + q * 256];
                bv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (use_b) bv[q] = reinterpret_cast<const f32x4*>(b)[i + q * 256];
                cols[q] = col4 * 4;
                col4 = advance(col4);
            }
#pragma unroll
            for (int q = 0; q < UQ; ++q) quad(i + q * 256, cols[q], av[q], bv[q]);
        }
        for (; i < hi; i += 256) {
            const f32x4 av = reinterpret_cast<const f32x4*>(a)[i];
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (use_b) bv = reinterpret_cast<const f32x4*>(b)[i];
            quad(i, col4 * 4, av, bv);
            col4 = advance(col4);
        }
    } else {
        for (long long i = lo + threadIdx.x; i < hi; i += 256) {
            const f32x4 av = reinterpret_cast<const f32x4*>(a)[i];
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (use_b) bv = reinterpret_cast<const f32x4*>(b)[i];
            quad(i, (int)((i * 4) % e.pitch), av, bv);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s_adv += __shfl_xor(s_adv, o, 64);
        s_fm += __shfl_xor(s_fm, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s_adv;
        red[1][threadIdx.x >> 6] = s_fm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* out = partials + ((size_t)blockIdx.y * kLossChunks + blockIdx.x) * 2;
        out[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        out[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// values[0] = sum_e c_adv[e] * S_adv[e], values[1] = sum_e c_fm[e] * S_fm[e], values[2] = lambda_adv * (values[0] + lambda_fm * values[1])
__global__ __launch_bounds__(256) void loss_reduce_kernel(const LossEntry* entries, int n, const float* partials, float* values, float lambda_adv,
                                                          float lambda_fm) {
    __shared__ float sa[256], sf[256];
    float ta = 0.f, tf = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) {  // (n <= 256 in practice: one entry per thread)
        float a = 0.f, f = 0.f;
        for (int c = 0; c < kLossChunks; ++c) {
            a += partials[((size_t)e * kLossChunks + c) * 2];
            f += partials[((size_t)e * kLossChunks + c) * 2 + 1];
        }
        ta += entries[e].c_adv * a;
        tf += entries[e].c_fm * f;
    }
    sa[threadIdx.x] = ta;
    sf[threadIdx.x] = tf;
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, f = 0.f;
        for (int i = 0; i < 256; ++i) {
            a += sa[i];
            f += sf[i];
        }
        values[0] = a;
        values[1] = f;
        values[2] = lambda_adv * (a + lambda_fm * f);
    }
}

}  // namespace hificar

// ------------------------------------------------------------------------------------------------
// Mel-spectrogram loss (articulatory/losses/mel_loss.py:16-166: torch.stft(center, hann) -> |.| -> mel filterbank -> log -> L1).  The DFT
// and the filterbank are GEMMs on the exact-fp32 conv kernels (window folded into the DFT matrix); these are the passes around them.
// ------------------------------------------------------------------------------------------------
namespace hificar {

struct FrameParams {
    const float* y;   // (B, T)
    float* A;         // [B * frames][N]
    long long total4; // B * frames * N / 4
    int T, N, hop, frames;
};

__device__ __forceinline__ int reflect_index(int i, int T) { return i < 0 ? -i : (i >= T ? 2 * (T - 1) - i : i); }

__global__ __launch_bounds__(256) void mel_frame_kernel(const FrameParams p) {  // A[(b, f)][n] = y_padded[b][f * hop + n], reflect padding N / 2
    const int n4 = p.N >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total4; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / n4), n = (int)(i - (long long)m * n4) * 4;
        const int b = m / p.frames, f = m - b * p.frames;
        f32x4 v;
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p.y[(size_t)b * p.T + reflect_index(f * p.hop + n + u - p.N / 2, p.T)];
        reinterpret_cast<f32x4*>(p.A)[i] = v;
    }
}

struct AmpParams {
    const float* S;   // [M][2 * nfp]: re at [0, nfp), im at [nfp, 2 nfp)
    float* amp;       // [M][nfp]
    const float* damp;  // backward: [M][nfp]
    float* dS;          // backward: [M][2 * nfp]
    long long total;  // M * nfp
    int nf, nfp;
    float eps;
};

__global__ __launch_bounds__(256) void mel_amp_kernel(const AmpParams p) {  // sqrt(clamp(re^2 + im^2, eps))
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / p.nfp), f = (int)(i - (long long)m * p.nfp);
        float a = 0.f;
        if (f < p.nf) {
            const float re = p.S[(size_t)m * 2 * p.nfp + f], im = p.S[(size_t)m * 2 * p.nfp + p.nfp + f];
            a = sqrtf(fmaxf(re * re + im * im, p.eps));
        }
        p.amp[i] = a;
    }
}

__global__ __launch_bounds__(256) void mel_amp_bwd_kernel(const AmpParams p) {  // d|S| -> d(re, im); the clamp passes no gradient below eps
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / p.nfp), f = (int)(i - (long long)m * p.nfp);
        float gre = 0.f, gim = 0.f;
        if (f < p.nf) {
            const float re = p.S[(size_t)m * 2 * p.nfp + f], im = p.S[(size_t)m * 2 * p.nfp + p.nfp + f];
            const float pw = re * re + im * im;
            if (pw > p.eps) {
                const float g = p.damp[i] / sqrtf(pw);
                gre = g * re;
                gim = g * im;
            }
        }
        p.dS[(size_t)m * 2 * p.nfp + f] = gre;
        p.dS[(size_t)m * 2 * p.nfp + p.nfp + f] = gim;
    }
}

struct MelLossParams {
    const float* mel_hat;  // [M][mp]
    const float* mel;      // [M][mp]
    float* dmel_hat;       // [M][mp] or nullptr
    float* partial;        // [gridDim.x]
    long long total;       // M * mp
    int nm, mp;
    float eps, log_scale;  // log_base: log(x) * log_scale
    float inv_numel;
};

__global__ __launch_bounds__(256) void mel_loss_kernel(const MelLossParams p) {  // L1 of log(clamp(mel, eps)), mean over (M, num_mels)
    __shared__ float red[4];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % p.mp);
        float g = 0.f;
        if (c < p.nm) {
            const float a = p.mel_hat[i], b = p.mel[i];
            const float la = logf(fmaxf(a, p.eps)) * p.log_scale, lb = logf(fmaxf(b, p.eps)) * p.log_scale;
            const float df = la - lb;
            s += fabsf(df);
            if (a > p.eps) g = (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * p.log_scale / a * p.inv_numel;
        }
        if (p.dmel_hat) p.dmel_hat[i] = g;
    }
    s = wg_sum256(s, red);
    if (threadIdx.x == 0) p.partial[blockIdx.x] = s * p.inv_numel;
}

struct OlaParams {
    const float* dA;  // [B * frames][N]
    float* dy;        // (B, T)
    long long total;  // B * T
    int T, N, hop, frames;
};

__device__ __forceinline__ float ola_at(const OlaParams& p, int b, int j) {  // sum over the frames that hold padded position j (offset by N / 2)
    const int q = j + p.N / 2;
    float s = 0.f;
    const int f_hi = min(q / p.hop, p.frames - 1);
    for (int f = max(0, (q - p.N + p.hop) / p.hop); f <= f_hi; ++f) {
        const int n = q - f * p.hop;
        if (n >= 0 && n < p.N) s += p.dA[((size_t)b * p.frames + f) * p.N + n];
    }
    return s;
}

__global__ __launch_bounds__(256) void mel_ola_kernel(const OlaParams p) {  // backward of mel_frame_kernel (reflect padding folded back)
    for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < p.total; i0 += (long long)gridDim.x * 256) {
        const int b = (int)(i0 / p.T), i = (int)(i0 - (long long)b * p.T);
        float s = ola_at(p, b, i);
        if (i > 0 && i <= p.N / 2) s += ola_at(p, b, -i);                                   // left mirror
        const int mr = 2 * (p.T - 1) - i;
        if (mr >= p.T && mr < p.T + p.N / 2) s += ola_at(p, b, mr);                         // right mirror
        p.dy[i0] = s;
    }
}

// Multi-resolution STFT loss (articulatory/losses/stft_loss.py:43-170), one resolution: on the magnitudes x = |STFT(y_hat)|, y = |STFT(y)|
//   spectral convergence  ||y - x||_F / ||y||_F        log STFT magnitude  mean |log y - log x|
// sums: per-workgroup partials of (sum (y - x)^2, sum y^2, sum |log y - log x|); final: the two values + the totals the gradient needs.
struct StftLossParams {
    const float* x;    // [M][nfp] magnitudes of the generated signal
    const float* y;    // ground truth
    float* partial;    // [blocks][3]
    float* sums;       // [4]: S1, S2, S3, -
    float* values;     // [2]: sc, mag
    const float* gw;   // backward: [2] upstream gradients of (sc, mag)
    float* dx;         // backward: [M][nfp]
    long long total;   // M * nfp
    int nf, nfp, blocks;
    float inv_numel;   // 1 / (M * nf)
};

__global__ __launch_bounds__(256) void stft_loss_sums_kernel(const StftLossParams p) {
    __shared__ float red[4];
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
        if ((int)(i % p.nfp) >= p.nf) continue;
        const float x = p.x[i], y = p.y[i];
        s1 += (y - x) * (y - x);
        s2 += y * y;
        s3 += fabsf(logf(y) - logf(x));
    }
    s1 = wg_sum256(s1, red);
    s2 = wg_sum256(s2, red);
    s3 = wg_sum256(s3, red);
    if (threadIdx.x == 0) {
        p.partial[blockIdx.x * 3] = s1;
        p.partial[blockIdx.x * 3 + 1] = s2;
        p.partial[blockIdx.x * 3 + 2] = s3;
    }
}

__global__ void stft_loss_final_kernel(const StftLossParams p) {  // one thread: fixed-order totals
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int b = 0; b < p.blocks; ++b) {
        s1 += p.partial[b * 3];
        s2 += p.partial[b * 3 + 1];
        s3 += p.partial[b * 3 + 2];
    }
    p.sums[0] = s1;
    p.sums[1] = s2;
    p.sums[2] = s3;
    p.values[0] = sqrtf(s1) / sqrtf(s2);
    p.values[1] = s3 * p.inv_numel;
}

__global__ __launch_bounds__(256) void stft_loss_grad_kernel(const StftLossParams p) {  // d(gw0 * sc + gw1 * mag) / dx
    const float s1 = p.sums[0], s2 = p.sums[1];
    const float csc = s1 > 0.f ? p.gw[0] / (sqrtf(s1) * sqrtf(s2)) : 0.f;
    const float cmag = p.gw[1] * p.inv_numel;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
        float g = 0.f;
        if ((int)(i % p.nfp) < p.nf) {
            const float x = p.x[i], y = p.y[i];
            const float dl = logf(x) - logf(y);
            g = csc * (x - y) + cmag * (dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f)) / x;
        }
        p.dx[i] = g;
    }
}

}  // namespace hificar
