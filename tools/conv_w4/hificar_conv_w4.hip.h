// Round 5: the dense exact-fp32 conv on a FOUR-wave workgroup (one wave per SIMD, the whole 512-entry register file per wave).
//
// conv_ws_body (hificar_kernels.hip.h) splits a workgroup into four MFMA waves and four loader waves; two waves share a SIMD, so a wave owns at most 256
// registers: 64 accumulators (MI = 4 row blocks x one 32-channel block) + operands is what fits, and a finished tile's epilogue (direct output, round 4)
// runs with the matrix pipe idle.  Here every wave does everything:
//   * it issues its quarter of the LDS-DMA of the NEXT (tile, chunk) item right behind the item barrier (global_load_lds_dwordx4 as inline asm: the
//     compiler's LDS-DMA tracking would put an s_waitcnt vmcnt in front of every following ds_read; the one wait a staged item needs is the explicit
//     vmcnt(0) in front of the item barrier, by which time the loads have long landed),
//   * runs the K loop on a register-blocked wave tile of MI x NB 32 x 32 blocks (NB = 2: 128 accumulators, 0.19 instead of 0.31 operand loads per MFMA),
//   * and (PIPE) keeps a finished tile as a second register set `pend` whose epilogue — residual loads, bias / residual adds, LeakyReLU, 16-byte stores —
//     is issued in pieces BETWEEN the MFMAs of the next tile's first two taps (same basic block: no branches inside a piece; rows past the end of a
//     sequence are dropped by the buffer instructions' range check instead of a divergent branch).
// Arithmetic order is conv_ws_body's, product for product and add for add: accumulators from zero over (chunk, tap, K slab, half, k), then
// (acc + bias) [* LeakyReLU'(mask)] + residual — results are bit-identical to the 8-wave kernels' (tests/test_gpu_parity.py).
//
// Reference: the convolutions of HiFiGANResidualBlock (articulatory/layers/residual_block.py:207-222) and the upsampling ConvTranspose1d
// (articulatory/models/hifigan.py:117-133, 224) — 99.5 % of the generator's MACs.
#pragma once

namespace hificar {

// one 1-KiB piece of an LDS-DMA: lane l's 16 bytes land at lds_addr + 16 l.  M0 carries the LDS address (wave-uniform).
__device__ __forceinline__ void dma16_nt(const char* src, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" : : "s"(lds_addr), "v"(src) : "memory");
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Pin of a K-slab step that also carries a piece of the previous tile's epilogue: behind MFMA I come its even share of the step's LDS reads, its
// VMEM reads (weight ring + residual rows), a few VALU instructions of the epilogue arithmetic and its share of the stores (VMEM write 0x40).
template <int I, int NDS, int NVR, int NVW, int NMFMA>
__device__ __forceinline__ void pin_epi_slot() {
    if constexpr (I < NMFMA) {
        constexpr int NMEM = NDS + NVR;
        constexpr int lo = NMEM * I / NMFMA, hi = NMEM * (I + 1) / NMFMA;
        constexpr int ds = (hi < NDS ? hi : NDS) - (lo < NDS ? lo : NDS);
        constexpr int vr = (hi - lo) - ds;
        constexpr int vw = NVW * (I + 1) / NMFMA - NVW * I / NMFMA;
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (ds > 0) __builtin_amdgcn_sched_group_barrier(0x100, ds, 0);
        if constexpr (vr > 0) __builtin_amdgcn_sched_group_barrier(0x020, vr, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // VALU: up to two per gap (the group takes what is there)
        if constexpr (vw > 0) __builtin_amdgcn_sched_group_barrier(0x040, vw, 0);
        pin_epi_slot<I + 1, NDS, NVR, NVW, NMFMA>();
    }
}
template <int NDS, int NVR, int NVW, int NMFMA>
__device__ __forceinline__ void pin_epi_step() {
#if HIFICAR_PIN == 2
    pin_epi_slot<0, NDS, NVR, NVW, NMFMA>();
#endif
}

// FULL = false: the launch only writes activated copies (`ys`; ResBlock conv1) — no residual, no fp32 output, no mask.  FULL = true: everything
// ConvParams offers (residual, mask, y and / or ys; a null pointer becomes an empty buffer range: its loads return zeros, its stores are dropped).
template <int MI, int WM, int WN, int NB, int NC16, bool PIPE, bool FULL>
__device__ __forceinline__ void conv_w4_body(const MultiConvParams& mp) {
    static_assert(WM * WN == 4, "four waves per workgroup, one per SIMD");
    static_assert(NB == 1 || NB == 2, "one or two channel blocks per wave");
    static_assert(NC16 == 2 || NC16 == 4, "chunk of 32 or 64 channels");
    constexpr int kFirstLoader = 1 << 20;  // (HIFICAR_STAMP: only wave 0 stamps)
    (void)kFirstLoader;
    constexpr int TM = WM * MI * 32;
    constexpr int CH = NC16 * 16;
    constexpr int RB = CH * 4;
    constexpr int SPR = CH / 4;
    constexpr int LOG_SPR = NC16 == 4 ? 4 : 3;
    constexpr int LOG_RPB = 4 - LOG_SPR;
    constexpr int NBLK = MI * NB;                 // 32 x 32 blocks per wave
    constexpr int NSLOT = 2 * NC16;               // epilogue slots: the slab steps of a tile's first two taps
    constexpr int BPS = (NBLK + NSLOT - 1) / NSLOT;  // blocks per slot
    extern __shared__ __attribute__((aligned(1024))) char smem_b[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int li = lane & 31;
    const int g = lane >> 5;
    const int nchunks = mp.p[0].cin / CH;
    const int tiles_per_branch = mp.ngroups * mp.nseq_tiles;
    const int buf_bytes = mp.buf_bytes;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_b;

    struct Tile {
        int b, z, ng, seq, t0;
    };
    auto decode = [&](int tile) {
        Tile T;
        const int bz = tile / tiles_per_branch;
        T.b = bz / mp.zrep;
        T.z = bz - T.b * mp.zrep;
        const int rem = tile - bz * tiles_per_branch;
        T.ng = rem / mp.nseq_tiles;
        const int m = rem - T.ng * mp.nseq_tiles;
        const int tps = mp.p[0].tiles_per_seq;
        T.seq = m / tps;
        T.t0 = (m - T.seq * tps) * TM;
        return T;
    };
    constexpr int TN = WN * NB * 32;
    // tile walk: as conv_ws_body (host schedule light -> heavy, or round-robin walked light first)
    const int sched_lo = mp.sched_start ? mp.sched_start[blockIdx.x] : 0;
    const int my_rounds = mp.sched_start ? mp.sched_start[blockIdx.x + 1] - sched_lo
                                         : (mp.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto tile_of = [&](int it) {
        int i = it;
        if ((blockIdx.x & 1) && my_rounds >= 3 && it >= my_rounds - 2) i = it == my_rounds - 1 ? my_rounds - 2 : my_rounds - 1;
        if (mp.sched_start) return mp.sched_tiles[sched_lo + i];
        if (mp.xcd_order) {
            const int x = (int)blockIdx.x & 7, l = (int)blockIdx.x >> 3;
            const int base = mp.total_tiles >> 3, rem = mp.total_tiles & 7;
            return x * base + min(x, rem) + l;
        }
        return (int)blockIdx.x + (my_rounds - 1 - i) * (int)gridDim.x;
    };
    auto nxt = [&](int i) {
        if (is_ragged(mp.p[0]))
            while (i < my_rounds) {
                const Tile T = decode(tile_of(i));
                if (T.t0 < seq_rows(mp.p[T.b], T.seq)) break;
                ++i;
            }
        return i;
    };

    // ---------------- staging: this wave's quarter of item (T, c) into ring buffer jj & 1 ----------------
    auto dma_item = [&](const Tile& T, int c, int jj) {
        const ConvParams& p = mp.p[T.b];
        const int R = TM + p.halo;
        const int ninstr = (R * SPR + 63) >> 6;
        const int Ls = p.x_rows ? p.x_rows : seq_rows(p, T.seq);
        const int row_bytes = p.x_row_bytes ? p.x_row_bytes : p.cin * 4;
        const char* const xs_z = p.xs + (size_t)T.z * mp.zs_x + (size_t)T.seq * (p.x_seq_bytes ? (size_t)p.x_seq_bytes : (size_t)p.L * p.cin * 4);
        const unsigned dst = lds0 + (unsigned)((jj & 1) * buf_bytes);
        const int c0b = c * CH * 2;
        for (int i = wave; i < ninstr; i += 4) {
            const int n = i * 64 + lane;
            const int r = n >> LOG_SPR;
            const int sl = (n & (SPR - 1)) ^ ((r >> LOG_RPB) & (SPR - 1));
            const int t = T.t0 + p.off_min + r;
            const char* src = p.zeros;
            if (r < R && t >= 0 && t < Ls) {
                const int ts = p.x_up > 1 ? (int)__umulhi((unsigned)t, p.x_up_rcp) : t;
                src = xs_z + (size_t)ts * row_bytes + 2 * c0b + sl * 16;
            }
            dma16_nt(src, dst + (unsigned)i * 1024u);
        }
    };

    // ---------------- weights: register ring, one tap ahead (as conv_ws_body) ----------------
    f32x16 acc[NB][MI];
    constexpr int NMF = 8 * MI * NB;
    f32x4 wr[NB][NC16][2];
    auto wstream = [&](const Tile& T) {
        const ConvParams& p = mp.p[T.b];
        const int nb = (T.ng * WN + wn) * NB;
        return reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.w16) + (size_t)T.z * mp.zs_w) + (size_t)(nb < p.n_blocks32 ? nb : 0) * p.ntaps * (p.cin / 16) * 128 + lane;
    };
    auto wstride = [&](const Tile& T) {
        const ConvParams& p = mp.p[T.b];
        return ((T.ng * WN + wn) * NB + 1 < p.n_blocks32) ? (long long)p.ntaps * (p.cin / 16) * 128 : 0LL;
    };
    const f32x4* wp = nullptr;
    const f32x4* wp2 = nullptr;
    int groups_left = 0;
    bool primed = false;
    auto prime = [&](const Tile& T) {
        wp = wstream(T);
#pragma unroll
        for (int u = 0; u < NC16; ++u) {
            wr[0][u][0] = wp[u * 128];
            wr[0][u][1] = wp[u * 128 + 64];
        }
        if constexpr (NB == 2) {
            wp2 = wp + wstride(T);
#pragma unroll
            for (int u = 0; u < NC16; ++u) {
                wr[1][u][0] = wp2[u * 128];
                wr[1][u][1] = wp2[u * 128 + 64];
            }
            wp2 += NC16 * 128;
        }
        wp += NC16 * 128;
        groups_left = nchunks * mp.p[T.b].ntaps - 1;
    };
    const int wave_row0 = wm * (MI * 32);

    auto addr_set = [&](int buf_off, int roff, int (&ad)[NC16][2]) {
        const int r0 = wave_row0 + li + roff;
        const int swz = (r0 >> LOG_RPB) & (SPR - 1);
        const int base = buf_off + r0 * RB;
#pragma unroll
        for (int u = 0; u < NC16; ++u) {
            ad[u][0] = base + (((4 * u + g) ^ swz) << 4);
            ad[u][1] = base + (((4 * u + 2 + g) ^ swz) << 4);
        }
    };
    auto load_x = [&](f32x4 (&xh)[MI], f32x4 (&xl)[MI], const int (&ad)[2]) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            xh[mi] = *reinterpret_cast<const f32x4*>(smem_b + ad[0] + mi * 32 * RB);
            xl[mi] = *reinterpret_cast<const f32x4*>(smem_b + ad[1] + mi * 32 * RB);
        }
    };
    auto mfma_step = [&](const f32x4 (&xh)[MI], const f32x4 (&xl)[MI], const f32x4 (&wh)[NB], const f32x4 (&wl)[NB]) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int q = 0; q < NB; ++q)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[q][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(wh[q][s4], xh[mi][s4], acc[q][mi], 0, 0, 0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int q = 0; q < NB; ++q)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[q][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[q][s4], xl[mi][s4], acc[q][mi], 0, 0, 0);
    };

    // ---------------- epilogue state ----------------
    // lane (li, g) owns time row li of each of its row blocks and, per register quad q, the four adjacent channels 8 q + 4 g + {0..3} of each channel block
    struct Out {
        // ranges of the tile's valid rows as (base address, bytes): wave-uniform scalars — the buffer descriptors are rebuilt from them right where they
        // are used (a descriptor carried across the tile loop ends up in vector registers and every access in a readfirstlane loop); bytes = 0 for a null
        // pointer: loads return zeros, stores are dropped
        unsigned long long y, ys, res, mask;
        unsigned nbytes, has_y, has_ys, has_res, has_mask;
        int off;      // byte offset of (row li of row block 0, channel 4 g of channel block 0) inside them
        int pitch32;  // bytes per 32 rows
        float slope_out, mask_slope;
    };
    auto uni64 = [](unsigned long long v) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    };
    auto rsrc_of = [&](unsigned long long base, unsigned bytes) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)uni64(base), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);  // raw buffer, 32-bit data format
    };
    auto out_of = [&](const Tile& T, const ConvParams& p, int nb) {
        Out o;
        const int rows_valid = max(min(TM, seq_rows(p, T.seq) - T.t0), 0);
        const size_t first = ((size_t)T.seq * p.L + T.t0) * p.cout_total;  // floats before the tile's first row
        o.nbytes = (unsigned)rows_valid * (unsigned)p.cout_total * 4u;
        o.y = (unsigned long long)(p.y ? p.y + (size_t)T.z * mp.zs_y + first : nullptr);
        o.ys = (unsigned long long)(p.ys ? reinterpret_cast<float*>(p.ys) + (size_t)T.z * mp.zs_y + first : nullptr);
        o.res = (unsigned long long)(p.res ? p.res + first : nullptr);
        o.mask = (unsigned long long)(p.mask_src ? p.mask_src + first : nullptr);
        o.has_y = p.y != nullptr;
        o.has_ys = p.ys != nullptr;
        o.has_res = p.res != nullptr;
        o.has_mask = p.mask_src != nullptr;
        o.off = ((wave_row0 + li) * p.cout_total + nb * 32 + 4 * g) * 4;
        o.pitch32 = 32 * p.cout_total * 4;
        o.slope_out = p.slope_out;
        o.mask_slope = p.mask_slope;
        return o;
    };
    auto ld16 = [](__amdgpu_buffer_rsrc_t r, int off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); };
    auto st16 = [](__amdgpu_buffer_rsrc_t r, int off, f32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0); };

    // direct epilogue of the tile just finished (the last tile of a workgroup, and every tile of a !PIPE kernel): out of `acc`
    auto epilogue_direct = [&](const Tile& T, const ConvParams& p, int nb) {
        const Out o = out_of(T, p, nb);
        const __amdgpu_buffer_rsrc_t r_y = rsrc_of(o.y, o.has_y ? o.nbytes : 0u), r_ys = rsrc_of(o.ys, o.has_ys ? o.nbytes : 0u);
        const __amdgpu_buffer_rsrc_t r_res = rsrc_of(o.res, o.has_res ? o.nbytes : 0u), r_mask = rsrc_of(o.mask, o.has_mask ? o.nbytes : 0u);
        const float* const bias_z = p.bias + (size_t)T.z * mp.zs_b;
#pragma unroll
        for (int b2 = 0; b2 < NB; ++b2) {
            if (nb + b2 >= p.n_blocks32) break;  // (partial channel group: the second block does not exist)
            f32x4 bv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = *reinterpret_cast<const f32x4*>(bias_z + (nb + b2) * 32 + 4 * g + 8 * q);
            constexpr int G2 = MI >= 2 ? 2 : 1;
#pragma unroll
            for (int m0 = 0; m0 < MI; m0 += G2) {
                f32x4 rs[G2][4], mk[G2][4];
                if constexpr (FULL) {
#pragma unroll
                    for (int mm = 0; mm < G2; ++mm)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int off = o.off + (m0 + mm) * o.pitch32 + b2 * 128 + q * 32;
                            rs[mm][q] = ld16(r_res, off);
                            if (o.has_mask) mk[mm][q] = ld16(r_mask, off);
                        }
                }
#pragma unroll
                for (int mm = 0; mm < G2; ++mm) {
                    const int mi = m0 + mm;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int off = o.off + mi * o.pitch32 + b2 * 128 + q * 32;
                        f32x4 v;
                        if constexpr (FULL) {
                            if (o.has_mask) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = (acc[b2][mi][4 * q + e] + bv[q][e]) * (mk[mm][q][e] > 0.f ? 1.f : o.mask_slope) + rs[mm][q][e];
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = (acc[b2][mi][4 * q + e] + bv[q][e]) + rs[mm][q][e];
                            }
                            st16(r_y, off, v);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = (acc[b2][mi][4 * q + e] + bv[q][e]) + 0.f;
                        }
                        f32x4 a;
#pragma unroll
                        for (int e = 0; e < 4; ++e) a[e] = fmaxf(v[e], v[e] * o.slope_out);
                        st16(r_ys, off, a);
                    }
                }
            }
        }
    };

    // pending tile (PIPE): biased accumulators + where they go; its epilogue runs inside the next tile's first two taps.  (PIPE launches carry no
    // mask_src — the host picks a !PIPE kernel for the data-gradient launches — and every branch has at least two taps.)
    // Slot S of NSLOT finishes blocks [S NBLK / NSLOT, (S + 1) NBLK / NSLOT); block k = channel block k / MI, row block k % MI.
    f32x16 pend[NB][MI];
    Out po;
    int po_off2 = 128;  // byte offset of the wave's second channel block; out of every range when the layer has no such block (partial channel group)
    bool have_pend = false;
    f32x4 prs[2][BPS][4];  // residual rows of slot S's blocks live in prs[S & 1], requested one slot ahead
    auto blk_off = [&](int k, int q) { return po.off + (k % MI) * po.pitch32 + (k / MI) * po_off2 + q * 32; };
    auto epi_request = [&](auto slot_c) {  // residual loads of slot S's blocks
        constexpr int S = decltype(slot_c)::value;
        if constexpr (FULL && S < NSLOT) {
            constexpr int lo = S * NBLK / NSLOT, hi = (S + 1) * NBLK / NSLOT;
#pragma unroll
            for (int k = lo; k < hi; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) prs[S & 1][k - lo][q] = ld16(rsrc_of(po.res, po.has_res ? po.nbytes : 0u), blk_off(k, q));
        }
    };
    auto epi_slot = [&](auto slot_c) {  // finish slot S's blocks; request slot S + 1's residual rows first
        constexpr int S = decltype(slot_c)::value;
        constexpr int lo = S * NBLK / NSLOT, hi = (S + 1) * NBLK / NSLOT;
        epi_request(std::integral_constant<int, S + 1>());
#pragma unroll
        for (int k = lo; k < hi; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int off = blk_off(k, q);
                f32x4 v;
                if constexpr (FULL) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = pend[k / MI][k % MI][4 * q + e] + prs[S & 1][k - lo][q][e];
                    st16(rsrc_of(po.y, po.has_y ? po.nbytes : 0u), off, v);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = pend[k / MI][k % MI][4 * q + e] + 0.f;
                }
                f32x4 a;
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = fmaxf(v[e], v[e] * po.slope_out);
                st16(rsrc_of(po.ys, po.has_ys ? po.nbytes : 0u), off, a);
            }
    };
    auto epi_flush = [&](auto first_c) {  // every remaining slot back to back (a pending tile whose wave sits the next tile out)
        constexpr int S0 = decltype(first_c)::value;
        if constexpr (S0 == 0) epi_request(std::integral_constant<int, 0>());
        if constexpr (NSLOT >= 4) {
            epi_slot(std::integral_constant<int, 0>());
            epi_slot(std::integral_constant<int, 1>());
            epi_slot(std::integral_constant<int, 2>());
            epi_slot(std::integral_constant<int, 3>());
        }
        if constexpr (NSLOT == 8) {
            epi_slot(std::integral_constant<int, 4>());
            epi_slot(std::integral_constant<int, 5>());
            epi_slot(std::integral_constant<int, 6>());
            epi_slot(std::integral_constant<int, 7>());
        }
    };

    // one K-slab step of the ring: take slab u's fragments, reload the registers for the following tap, multiply
    auto slab_step = [&](const f32x4 (&xh)[MI], const f32x4 (&xl)[MI], int u) {
        f32x4 wh[NB], wl[NB];
        wh[0] = wr[0][u][0];
        wl[0] = wr[0][u][1];
        wr[0][u][0] = wp[u * 128];
        wr[0][u][1] = wp[u * 128 + 64];
        if constexpr (NB == 2) {
            wh[1] = wr[1][u][0];
            wl[1] = wr[1][u][1];
            wr[1][u][0] = wp2[u * 128];
            wr[1][u][1] = wp2[u * 128 + 64];
        }
        mfma_step(xh, xl, wh, wl);
    };
    // pins (sched_group_barrier; see pin_slab_step): a plain step spreads its 2 MI LDS reads and 2 NB weight loads behind its leading MFMAs; a step
    // that carries an epilogue slot additionally gets that slot's residual loads, VALU work and stores dealt out one MFMA gap at a time
    auto pin_plain = [&]() { pin_slab_step<2 * MI, 2 * NB, NMF>(); };

    int j = 0;
    HIFICAR_STAMP(0);
    int it = nxt(0);
    if (it >= my_rounds) return;  // (every wave of the workgroup takes the same decision: no barrier is left waiting)
    Tile T = decode(tile_of(it));
    dma_item(T, 0, 0);
    for (;;) {
        const int itn = nxt(it + 1);
        const bool more = itn < my_rounds;
        const Tile Tn = decode(tile_of(more ? itn : it));
        const ConvParams& p = mp.p[T.b];
        const int nb = (T.ng * WN + wn) * NB;
        const bool active = nb < p.n_blocks32;
        const int phase = active ? nb / p.nb32_per_phase : 0;
        const int roff0 = __builtin_amdgcn_readfirstlane(p.tap_off0[phase] - p.off_min);
        const int tap_step = p.tap_step;
        const int ntaps = p.ntaps;
        const f32x4* wp_next = wstream(Tn);
        const long long wst_next = NB == 2 ? wstride(Tn) : 0LL;
        const int groups_next = nchunks * mp.p[Tn.b].ntaps;
        if (active && !primed) prime(T);
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][mi][r] = 0.f;
        if constexpr (PIPE) {
            if (have_pend) {
                if (active && ntaps >= 2) {
                    epi_request(std::integral_constant<int, 0>());
                } else {  // this wave sits the tile out (partial channel group), or a one-tap tile: nothing to interleave with
                    epi_flush(std::integral_constant<int, 0>());
                    have_pend = false;
                }
            }
        }

        for (int c = 0; c < nchunks; ++c, ++j) {
            HIFICAR_STAMP(1 + 3 * j);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of item j has landed (and with it the weight ring's loads)
            __syncthreads();                                   // ... everybody's has; nobody reads item j-1's buffer any more
            HIFICAR_STAMP(2 + 3 * j);
            if (c + 1 < nchunks) dma_item(T, c + 1, j + 1);
            else if (more) dma_item(Tn, 0, j + 1);
            if (!active) continue;
            const int buf_off = (j & 1) * buf_bytes;
            int ad[NC16][2];
            addr_set(buf_off, roff0, ad);
            f32x4 x0h[MI], x0l[MI], x1h[MI], x1l[MI];
            load_x(x0h, x0l, ad[0]);
            // TI = 0 / 1: the tap carries the pending tile's epilogue slots TI * NC16 .. ; -1: plain
            auto tap_body = [&](int t, auto ti_c) {
                constexpr int TI = decltype(ti_c)::value;
                const bool last_tap = t + 1 == ntaps;
                int adn[NC16][2];
                addr_set(buf_off, roff0 + (last_tap ? t : t + 1) * tap_step, adn);
                if (groups_left == 0) {
                    wp = wp_next;
                    if constexpr (NB == 2) wp2 = wp_next + wst_next;
                    groups_left = groups_next;
                }
                --groups_left;
#pragma unroll
                for (int u = 0; u < NC16; u += 2) {
                    load_x(x1h, x1l, ad[u + 1]);
                    slab_step(x0h, x0l, u);
                    if constexpr (TI >= 0) {
                        if (u == 0) epi_slot(std::integral_constant<int, TI * NC16 + 0>());
                        else epi_slot(std::integral_constant<int, TI * NC16 + (NC16 > 2 ? 2 : 0)>());
                        pin_epi_step<2 * MI, 2 * NB + (FULL ? 4 * BPS : 0), (FULL ? 8 : 4) * BPS, NMF>();
                    } else {
                        pin_plain();
                    }
                    if (u + 2 < NC16) load_x(x0h, x0l, ad[u + 2]);
                    else load_x(x0h, x0l, adn[0]);
                    slab_step(x1h, x1l, u + 1);
                    if constexpr (TI >= 0) {
                        if (u == 0) epi_slot(std::integral_constant<int, TI * NC16 + 1>());
                        else epi_slot(std::integral_constant<int, TI * NC16 + (NC16 > 2 ? 3 : 1)>());
                        pin_epi_step<2 * MI, 2 * NB + (FULL ? 4 * BPS : 0), (FULL ? 8 : 4) * BPS, NMF>();
                    } else {
                        pin_plain();
                    }
                }
                wp += NC16 * 128;
                if constexpr (NB == 2) wp2 += NC16 * 128;
#pragma unroll
                for (int u = 0; u < NC16; ++u) {
                    ad[u][0] = adn[u][0];
                    ad[u][1] = adn[u][1];
                }
            };
            int t = 0;
            if constexpr (PIPE) {
                if (c == 0 && have_pend) {  // (ntaps >= 2: checked at the tile's start)
                    tap_body(0, std::integral_constant<int, 0>());
                    tap_body(1, std::integral_constant<int, 1>());
                    t = 2;
                    have_pend = false;
                }
            }
            for (; t < ntaps; ++t) tap_body(t, std::integral_constant<int, -1>());
        }
        HIFICAR_STAMP(3 * j);
        primed = active;
        if (active) {
            if (PIPE && more) {
                // keep the tile as (acc + bias); its epilogue rides on the next tile's first two taps
                const float* const bias_z = p.bias + (size_t)T.z * mp.zs_b;
                po = out_of(T, p, nb);
                po_off2 = nb + 1 < p.n_blocks32 ? 128 : 0x40000000;
#pragma unroll
                for (int b2 = 0; b2 < NB; ++b2) {
                    const int nbb = nb + b2 < p.n_blocks32 ? nb + b2 : nb;  // (a missing second block: its columns are never stored — see below)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_z + nbb * 32 + 4 * g + 8 * q);
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                            for (int e = 0; e < 4; ++e) pend[b2][mi][4 * q + e] = acc[b2][mi][4 * q + e] + bv[e];
                    }
                }
                have_pend = true;
            } else {
                epilogue_direct(T, p, nb);
            }
        }
        if (!more) break;
        it = itn;
        T = Tn;
    }
    HIFICAR_STAMP(62);
    HIFICAR_STAMP(63);
}

template <int MI, int WM, int WN, int NB, int NC16, bool PIPE, bool FULL>
__global__ __launch_bounds__(256) void conv_f32w4_kernel(const MultiConvParams mp) {
    conv_w4_body<MI, WM, WN, NB, NC16, PIPE, FULL>(mp);
}

}  // namespace hificar
