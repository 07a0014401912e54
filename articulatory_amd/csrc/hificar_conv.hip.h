// HIP kernels for the HiFi-GAN / HiFi-CAR generator forward pass on gfx950 (CDNA4, wave64).
//
// Internal activation layout is CHANNELS-LAST: (sequence, time, channel) with the channel axis contiguous.
// The reference's (B, C, T) layout exists only at the C-ABI boundary (feature input, waveform output), which
// `front_kernel` / `output_conv_kernel` convert on the fly.  Channels-last makes every halo a whole-row affair
// (zero rows outside [0, L) reproduce the reference's per-conv zero padding exactly), keeps global accesses
// 16-byte aligned for any tap offset, and puts the GEMM reduction axis (input channels) contiguous for MFMA
// operand fragments.
//
// Every Conv1d and every ConvTranspose1d of the generator is the same implicit GEMM
//     D[t, co] = sum_{tap} sum_{ci} act(X)[t + off(tap), ci] * W[tap][ci][co]        (K = taps x input channels)
// A ConvTranspose1d with K = 2*stride is that contraction with N = stride*Cout "virtual" channels (phase-major) and
// per-phase tap lists, because out[(q*s + r), co] in channels-last memory IS row q, column r*Cout + co.
//
// Two arithmetics on one persistent, wave-specialised, LDS-DMA-staged kernel body:
//   conv_f32do_kernel          exact fp32 products (v_mfma_f32_32x32x2_f32), plain fp32 rows, tiles stored straight from the accumulators
//   conv_bf16x3(nb)_kernel     fp32 operands split hi+lo bf16, 3 x v_mfma_f32_32x32x16_bf16 per K slab; "split rows"
//   conv_sk_{f32,bf16x3}_kernel  split-K forms of both for launches with few tiles
//   conv_pair_{f32,bf16x3}_kernel  fuse conv1 -> conv2 of a ResBlock layer pair at C <= 64
// plus front_kernel (PastFCEncoder + input assembly), mrf_split_kernel (MRF mean + split), output_conv_kernel.
// The A/B forms of rounds 2-5 that lost their measurements (chained stage launches, fp32 register blocking, the out-buffer fp32 dense form,
// overlapped / 16-byte epilogues, split output pass) live on only in tools/r05_kernels/, with the evidence under profiles/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace hificar {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kMaxPhase = 8;
constexpr int kMaxTaps = 16;

// One conv layer ("branch") of a launch.  Activations travel between layers already activated, 4*C bytes per row:
//   bf16x3 arithmetic: "split rows" [hi: C bf16 | lo: C bf16] of LeakyReLU(x);  exact-fp32 arithmetic: C floats of LeakyReLU(x).
struct ConvParams {
    const bf16x8* w16;  // MFMA weight fragments [n_block32][chunk][tap][c16][hi|lo or half][lane], 16 B each (bf16 pairs or fp32)
    const float* bias;  // [cout_total] (never null; zeros when the layer has no bias)
    const float* res;   // fp32 residual, same layout as y, or null
    const float* mask_src;  // backward (dgrad) launches: rows in y's layout; the result is scaled by LeakyReLU'(mask_src) = (m > 0 ? 1 : mask_slope)
                            // BEFORE the residual is added (dx = dx_skip + act'(x) * dgrad); null in every forward launch
    float mask_slope;
    float* y;           // fp32 output of the layer itself (pre-activation), or null when only the activated copy is consumed
    const float* xf;    // pair kernel only: fp32 PRE-activation input rows (C floats per row); LeakyReLU(slope_in) + split are
                        // then applied while staging and xs is null (narrow stages: the producer writes no activated copy)
    float slope_in;
    int act_in;         // exact-fp32 conv kernels (conv_ws_body): n >= 1 = the input is the MEAN of n PRE-activation fp32 streams xs, xs_more[0 .. n - 2]
                        // (same layout), summed in that order and divided by n — the MRF mean of hifigan.py:226-230 folded into its consumer: the
                        // loader waves stage the rows through registers (n global loads, the sum, the division, LeakyReLU(slope_in), ds_write per
                        // 16 bytes) instead of the LDS-DMA, so no kernel writes the mean or its activated copy.  0: xs is activated already (LDS-DMA)
    const char* xs;     // input rows, already activated (and split) by their producer
    const char* xs_more[3];
    char* ys;           // activated (and split) copy of the output for the consumer conv: LeakyReLU(out, slope_out), or null
    const char* zeros;  // >= 16 bytes of zeros (source of padding rows for the LDS DMA)
    float slope_out;
    int cout_real;      // channels per real output row (== cout_total except for the polyphase ConvTranspose1d)
    int L;              // rows (time steps) per sequence, input rows == output rows
    int tiles_per_seq;  // ceil(L / TM)
    int cin;            // padded input channels == row pitch of xs in elements
    int cout_total;     // row pitch of y / res / bias length
    int n_blocks32;     // number of 32-wide output blocks (cout_total / 32)
    int nb32_per_phase;
    int ntaps;          // taps per phase (same for all phases; missing taps have zero weights)
    int off_min;        // min over all tap offsets (<= 0)
    int halo;           // off_max - off_min
    // tap t of phase r reads input row  t_out + tap_off0[r] + t * tap_step  (an arithmetic progression for both
    // Conv1d: -padding + t*dilation, and the polyphase ConvTranspose1d: floor((r+p)/s) - t); no per-tap table
    // lookups in the K loop (a memory lookup there would drain the weight prefetch queue with vmcnt(0)).
    int tap_step;
    int tap_off0[kMaxPhase];
    // ragged batches: utterance b is seq_len[b] frames long; this launch covers frames [len_f0, len_f0 + len_max) at
    // len_mul rows per frame, so sequence b has clamp(seq_len[b] - len_f0, 0, len_max) * len_mul valid rows (<= L): rows
    // beyond them read as zero padding and are never written.  seq_len == null: every sequence has L rows.
    // len_const >= 0 (seq_len null): every sequence has len_const frames (a launch over a bucketed length).
    const int* seq_len;
    int len_const;
    int len_f0, len_max, len_mul;
    // input row addressing (conv_ws_body's staging only): row t of sequence s starts at xs + s * x_seq_bytes + t * x_row_bytes.  The
    // defaults (0) mean packed rows: x_row_bytes = cin * 4, x_seq_bytes = L * cin * 4.  A row pitch SHORTER than the row (overlapping
    // rows) makes the staged rows sliding windows of one flat buffer: a strided conv in GEMM form reads its im2col rows
    // A[t] = x_flat[t * stride * cin_g .. + k * cin_g) straight from the activations (hificar_disc.hip.inc), no im2col copy.
    long long x_seq_bytes;
    int x_row_bytes;
    // input rows that exist per sequence when that differs from the output rows (0: the same, L / seq_rows): a strided conv run as a
    // multi-tap conv over rows of `stride` input positions (hificar_disc.hip.inc, polyphase-input form) reads ntaps - 1 rows past its
    // last output row, and its data gradient writes more rows than it reads.  Rows in [0, x_rows) are staged, the rest read as zeros.
    int x_rows;
    // nearest-neighbour upsampling of the input rows in front of the conv (torch.nn.Upsample(scale_factor = x_up) of a GBlock,
    // articulatory/layers/pytorch_layers.py:48-56): staged row t reads input row t / x_up, so the upsampled tensor never exists in HBM.
    // 0 / 1: none.  The sequence pitch of the (shorter) input then comes from x_seq_bytes.  x_up_rcp = floor(2^32 / x_up) + 1 (host):
    // t / x_up == umulhi(t, x_up_rcp) for 0 <= t < 2^32 / x_up — two instructions in the staging loop instead of a division.
    int x_up;
    unsigned x_up_rcp;
};

// A wave-uniform int the HOST wrote before the launch (tile schedules, sequence lengths), read through the scalar cache.  A plain load of it compiles to
// a VECTOR load followed by s_waitcnt vmcnt(0) — the kernels store to global memory, so hipcc will not use the scalar cache on its own — and that wait
// also sits out every store the wave has in flight: between two tiles of the direct-output conv kernels, the whole epilogue (2.8 k cycles per tile).
__device__ __forceinline__ int scalar_load_i32(const int* p) {
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// two consecutive ints (a workgroup's [start, end) of the tile schedule) in one request
__device__ __forceinline__ void scalar_load_2i32(const int* p, int& a, int& b) {
    long long v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    a = (int)v;
    b = (int)(v >> 32);
}

__device__ __forceinline__ bool is_ragged(const ConvParams& p) { return p.seq_len != nullptr || p.len_const >= 0; }

// valid rows of sequence `seq` (wave-uniform: one scalar load)
__device__ __forceinline__ int seq_rows(const ConvParams& p, int seq) {
    if (!is_ragged(p)) return p.L;
    const int n = (p.seq_len ? scalar_load_i32(p.seq_len + __builtin_amdgcn_readfirstlane(seq)) : p.len_const) - p.len_f0;
    return min(max(n, 0), p.len_max) * p.len_mul;
}

struct MultiConvParams {
    ConvParams p[3];
    // persistent-tile bookkeeping (wave-specialised kernel): tiles are numbered branch-major (heaviest branch
    // first), then channel group, then (sequence, time tile); workgroup w takes tiles w, w + gridDim.x, ...
    int n_branches;
    int nseq_tiles;   // sequences * tiles_per_seq
    int ngroups;      // ceil(n_blocks32 / WN)
    int total_tiles;  // n_branches * ngroups * nseq_tiles
    int buf_bytes;    // bytes of one LDS staging buffer (sized for the widest halo among the branches)
    // host-computed schedule (longest-processing-time assignment of tiles to workgroups, each list ordered light
    // first): workgroup w walks sched_tiles[sched_start[w] .. sched_start[w+1]).  Null: round-robin w, w+G, ...
    const int* sched_start;
    const int* sched_tiles;
    // every branch replicated zrep times (the groups of a grouped conv, hificar_disc.hip.inc): replica z reads xs + z * zs_x bytes and
    // w16 + z * zs_w bytes, bias + z * zs_b floats, writes y / ys + z * zs_y floats.  Tiles are numbered (branch, replica)-major.
    int zrep;
    long long zs_x, zs_w, zs_y, zs_b;
    unsigned long long* trace;  // dev tool only (tools/conv_bench.hip, -DHIFICAR_TRACE): per-workgroup timeline
    int xcd_order;  // 1 (gridDim.x == total_tiles, no schedule): XCD-contiguous tile order, see tile_of
    int stage_cached;  // 1: the staging loads are ordinary cached loads — every input row is staged by MORE than two workgroups (one per channel group
                       // of a wide layer: the ten groups of upsampler 0, the eight of a 1024-wide discriminator GEMM), so the re-reads should hit
                       // the L2.  0: non-temporal (a row is staged by one or two CUs only: +2.3 % end to end on the ResBlock launches)
};


#ifdef HIFICAR_TRACE
// (dev builds only) s_memtime is a per-XCD shader-clock counter; the first and the last stamp of a workgroup's wave 0 additionally record
// s_memrealtime (100 MHz, one time base for the whole device) so that a launch's span and the gap to the next launch can be measured
__device__ unsigned long long g_trace_realtime[2][1024];
__device__ unsigned long long g_trace_rt_slots[4][64];  // s_memrealtime of every stamp of wave 0 of workgroups 0 .. 3
#define HIFICAR_STAMP(slot)                                                                            \
    do {                                                                                               \
        if (mp.trace && lane == 0 && (wave == 0 || wave == kFirstLoader) && (slot) < 64)               \
            mp.trace[((size_t)blockIdx.x * 2 + (wave ? 1 : 0)) * 64 + (slot)] = __builtin_amdgcn_s_memtime(); \
        if (mp.trace && lane == 0 && wave == 0 && ((slot) == 0 || (slot) == 63) && blockIdx.x < 1024)  \
            g_trace_realtime[(slot) == 63][blockIdx.x] = __builtin_amdgcn_s_memrealtime();             \
        if (mp.trace && lane == 0 && wave == 0 && (slot) < 64 && blockIdx.x < 4)                       \
            g_trace_rt_slots[blockIdx.x][(slot)] = __builtin_amdgcn_s_memrealtime();                   \
    } while (0)
#else
#define HIFICAR_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ float lrelu(float v, float slope) { return v >= 0.f ? v : v * slope; }


// Software-pipeline pin for one K-slab step of the conv kernels.  The source issues the NEXT slab's LDS fragment reads (and the weight
// fragment loads two taps ahead) before the CURRENT slab's MFMAs, but hipcc's scheduler sinks every ds_read to just above its first use
// and waits on it at once (ds_read ; s_waitcnt lgkmcnt ; v_mfma ...), which exposes the LDS latency once per slab: 17 % of the fp32 K loop
// and ~40 % of the bf16x3 one (measured: s_memtime timeline / rocprof).  The pin: every MFMA of a slab step is followed by
// its even share of the step's memory instructions (sched_group_barrier: MFMA 0x8, DS read 0x100, VMEM read 0x20).
template <int I, int NDS, int NVMEM, int NMFMA>
__device__ __forceinline__ void pin_slab_slot() {
    if constexpr (I < NMFMA) {
        // MFMA I, then this slot's even share of the memory instructions (LDS reads first, then the weight loads)
        constexpr int NMEM = NDS + NVMEM;
        constexpr int lo = NMEM * I / NMFMA, hi = NMEM * (I + 1) / NMFMA;
        constexpr int ds = (hi < NDS ? hi : NDS) - (lo < NDS ? lo : NDS);
        constexpr int vm = (hi - lo) - ds;
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (ds > 0) __builtin_amdgcn_sched_group_barrier(0x100, ds, 0);
        if constexpr (vm > 0) __builtin_amdgcn_sched_group_barrier(0x020, vm, 0);
        pin_slab_slot<I + 1, NDS, NVMEM, NMFMA>();
    }
}
template <int NDS, int NVMEM, int NMFMA>
__device__ __forceinline__ void pin_slab_step() {
    pin_slab_slot<0, NDS, NVMEM, NMFMA>();
}
// An MFMA wave's barrier: it orders LDS (staged items in, out-buffer hand-over out), never the wave's own global loads / stores — __syncthreads() would
// also wait (vmcnt(0)) for the weight-ring loads just requested and for every store of the previous tile's epilogue.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution with fp32 operands split into bf16 hi + lo ("bf16x3"):
//     x*w ~= x_hi*w_hi + x_hi*w_lo + x_lo*w_hi,   x_hi = bf16(x), x_lo = bf16(x - x_hi)
// three v_mfma_f32_32x32x16_bf16 per 16-channel K slab, fp32 accumulate.  Each operand carries a 16-bit
// significand (relative product error ~2^-16; measured end-to-end error vs the fp32 oracle ~1.5e-5 of max|y|,
// two orders inside the 1e-3 parity bar) at 16/3 = 5.3x the fp32-MFMA rate.
//
// Activations travel between layers as "split rows" [hi: C bf16 | lo: C bf16] of LeakyReLU(x): the PRODUCER's
// epilogue activates and splits once, so a consumer stages its input with pure LDS-DMA
// (global_load_lds_dwordx4: no VGPRs, no VALU, zero rows from a zero page).  The fp32 value is written
// next to it only where a residual add or the MRF mean needs it.
//
// Persistent + wave-specialised: 8 waves per workgroup, one workgroup per CU, each workgroup walks a static
// list of output tiles (ids w, w+G, ...; ordered heaviest ResBlock branch first so every workgroup gets the same
// mix of kernel sizes).  Waves 0-3 ("MFMA waves") only read LDS and issue MFMAs; when a tile is finished they
// drop the raw accumulators into an LDS out-buffer and move on.  Waves 4-7 ("loaders") issue the DMA for the
// NEXT (tile, channel chunk) item into a 2-deep LDS ring — across tile boundaries too — and run the whole
// output pass of the PREVIOUS tile (bias, residual, fp32 store, LeakyReLU + split store) out of that buffer
// with row-contiguous accesses, so neither staging nor the epilogue is ever on the matrix pipe's critical path.
// One barrier per item; every launch uses >= 2 chunks per tile so that the out-buffer hand-off cannot race.
//   tile = TM = WM*MI*32 time rows x TN = WN*32 channels; a wave owns MI 32x32 tiles stacked in time.
//   LDS item = rows [t0+off_min, t0+TM+off_max) x CH channels, row = [hi CH | lo CH] = CH/4 16-byte slots,
//   slot index XOR-swizzled by the row so that the 16 lanes of a ds_read_b128 group hit 16 distinct slots
//   of the 256-byte bank row (LDS-DMA forbids padding: the destination is lane-linear).
//   MFMA: D^T = W * X^T — weights are the A operand (pre-packed in lane order, streamed from L2 one tap
//   ahead through a register ring whose pointer jumps to the next tile's stream during a tile's last tap),
//   activations the B operand (lane l -> time row l&31, channels 8*(l>>5)+j = 16 contiguous bytes), software-
//   pipelined one K slab ahead.  So a lane ends with 4 adjacent channels per register quad: 16-byte stores (out-buffer forms; the
//   direct-output form swaps the operands: see DOUT below).
// ------------------------------------------------------------------------------------------------
//
// F32 = true is the exact-fp32 arithmetic on the same skeleton: rows are plain fp32 LeakyReLU(x) (same 4*C bytes), a
// 16-channel K slab is two 8-channel halves (lane (li, g) holds channels 8v + 4g + {0..3} of its time row: one
// ds_read_b128 = four v_mfma_f32_32x32x2_f32 steps), weights are fp32 fragments in the same [slab][half][lane][16 B]
// order.  8 MFMAs of 64 cycles per slab and accumulator instead of 3 of 32: staging, weight stream and output pass
// vanish behind the matrix pipe.
//
// WM * WN = 4 or 8 MFMA waves.  8 (a 768-thread workgroup: two MFMA waves and one loader wave per SIMD) is supported by the body but
// not instantiated: measured 3-5 % slower than 4 waves on the same tile in both arithmetics (tools/conv_bench.hip has the runs) —
// the ~13 % of K-loop cycles without MFMA issue are the same with one or two MFMA waves per SIMD, i.e. not an issue-gap problem.
//
// KS = 4 is the split-K form for launches with few tiles (small batches: a chunk of one utterance is 25-125 rows per stage): the tile
// is MI*32 rows x 32 channels and the four MFMA waves split its K loop — wave ks takes the (tap, slab) steps s = ks (mod 4) — so a
// 32-channel block's dependent MFMA chain is a quarter as long and four times as many workgroups have work.  Each wave leaves a partial
// accumulator in the out-buffer; the output pass sums them in the fixed order (p0 + p1) + (p2 + p3): deterministic, but a different
// rounding order than the dense form (results agree to fp32 rounding, not bit for bit).
//
// NB = 2 is the register-blocked wave tile: an MFMA wave owns NB adjacent 32-channel blocks x MI row blocks (NB * MI accumulators), so one
// activation fragment read feeds NB times the MFMAs and one weight fragment MI of them: 2 (MI + NB) 16-byte loads per 8 MI NB MFMAs per K slab
// (fp32) — MI = 2, NB = 2: 0.25 loads per MFMA against 0.31 at MI = 4, NB = 1 on the same 64 accumulator registers and the same LDS budget
// (tile = WM*MI*32 rows x WN*NB*32 channels).  Every 16-byte load next to fp32 MFMAs costs ~22 cycles of matrix-pipe time.
//
// DOUT = true (every dense exact-fp32 launch): the MFMA waves write a finished tile straight from their accumulators
// (bias, residual, LeakyReLU, stores) — no LDS out-buffer, no output pass in the loader waves, which then only stage.  Trades the loaders' share of the SIMDs' issue
// slots during the K loop (and the waits at the out-buffer hand-over barriers) for an epilogue the matrix pipe idles through: +1.5 % end to end (37.41 -> 37.96 M).
// Round 5: in this form the operands are SWAPPED — D = X * W^T, activations as the A operand, same products in the same order — so that a lane owns ONE channel and
// 16 rows per 32 x 32 block and a 4-byte wave-store writes two whole 128-byte rows (kRowMajorAcc / epilogue_rm below; 38.4 -> 39.6 M with the waits around it removed).
template <int MI, int WM, int WN, int NC16, bool F32, int KS = 1, int NB = 1, bool DOUT = false>
__device__ __forceinline__ void conv_ws_body(const MultiConvParams& mp) {
    static_assert(!DOUT || (F32 && KS == 1 && NB == 1), "direct output: exact fp32, dense form, one channel block per wave");
    static_assert(NB == 1 || (NB == 2 && KS == 1), "one or two channel blocks per MFMA wave");
    static_assert(KS == 1 || (KS == 4 && WM == 1 && WN == 1), "split-K: four waves share one 32-channel block");
    static_assert(KS == 4 || WM * WN == 4 || WM * WN == 8, "4 or 8 MFMA waves per workgroup");
    constexpr int NW = KS == 4 ? 4 : WM * WN;  // MFMA waves; waves NW .. NW+3 are the loaders
    // Direct output with the accumulators transposed (round 5): lane (li, g) owns channel li of its block and rows 8 q + 4 g + e (register 4 q + e),
    // so one 4-byte wave-store writes two whole 128-byte rows of the block where a 16-byte form touches 32 rows with 32 bytes each (one request
    // per lane in the CU's vector-memory path: ~70 cycles per wave-store, 4.5 k cycles per tile and CU with the matrix pipe idle).
    constexpr bool kPrimeOnce = DOUT;  // the weight ring primed once in front of the tile loop (see there)
    constexpr bool kRowMajorAcc = DOUT;
    constexpr int NTHR = (NW + 4) * 64;
    constexpr int kFirstLoader = NW;
    (void)kFirstLoader;
    static_assert(NC16 == 1 || NC16 == 2 || NC16 == 4, "chunk of 16, 32 or 64 channels");
    constexpr int TM = WM * MI * 32;
    constexpr int CH = NC16 * 16;
    constexpr int RB = CH * 4;            // bytes per LDS row
    constexpr int SPR = CH / 4;           // 16-byte slots per row (4, 8, 16)
    constexpr int LOG_SPR = NC16 == 4 ? 4 : NC16 == 2 ? 3 : 2;
    constexpr int LOG_RPB = 4 - LOG_SPR;  // log2(rows per 256-byte bank row)
    extern __shared__ __attribute__((aligned(1024))) char smem_b[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= NW;
    const int cw = loader ? 0 : wave;
    const int wm = KS == 4 ? 0 : cw / WN;
    const int wn = KS == 4 ? 0 : cw % WN;
    const int li = lane & 31;
    const int g = lane >> 5;
    const int nchunks = mp.p[0].cin / CH;  // identical for all branches of a launch
    const int tiles_per_branch = mp.ngroups * mp.nseq_tiles;
    const int buf_bytes = mp.buf_bytes;

    struct Tile {
        int b, z, ng, seq, t0;
    };
    auto decode = [&](int tile) {
        Tile T;  // (unsigned divisions: half the scalar instructions of signed ones, and this runs between two tiles of every workgroup)
        const unsigned ut = (unsigned)tile, tpb = (unsigned)tiles_per_branch, zr = (unsigned)mp.zrep, nst = (unsigned)mp.nseq_tiles;
        const unsigned bz = ut / tpb;
        const unsigned b = bz / zr;
        T.b = (int)b;
        T.z = (int)(bz - b * zr);
        const unsigned rem = ut - bz * tpb;
        const unsigned ng = rem / nst;
        T.ng = (int)ng;
        const unsigned m = rem - ng * nst;
        const unsigned tps = (unsigned)mp.p[0].tiles_per_seq;
        const unsigned seq = m / tps;
        T.seq = (int)seq;
        T.t0 = (int)((m - seq * tps) * (unsigned)TM);
        return T;
    };

    constexpr int TN = WN * NB * 32;
    constexpr int OP = TN + 4;  // out-buffer row pitch (floats)
    // Tile walk.  With a host schedule: the tiles assigned to this workgroup, already ordered light -> heavy.
    // Without: tiles w, w+G, w+2G, ... (rounds run heavy -> light because tile ids are heaviest-branch-major), walked
    // LIGHT FIRST: the loader waves write a finished tile out while the MFMA waves compute the next one, and that only
    // hides completely behind a tile at least as heavy.  Odd workgroups swap their last two tiles so that
    // neighbouring CUs are not in the same phase all the time (synchronised DMA / output bursts cost ~15 % here).
    int sched_lo = 0, sched_hi = 0;
    if (mp.sched_start) scalar_load_2i32(mp.sched_start + blockIdx.x, sched_lo, sched_hi);
    const int my_rounds = mp.sched_start ? sched_hi - sched_lo
                                         : (mp.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto tile_of = [&](int it) {
        int i = it;
        if ((blockIdx.x & 1) && my_rounds >= 3 && it >= my_rounds - 2) i = it == my_rounds - 1 ? my_rounds - 2 : my_rounds - 1;
        if (mp.sched_start) return scalar_load_i32(mp.sched_tiles + sched_lo + i);
        if (mp.xcd_order) {
            // one tile per workgroup, weights outweigh activations (small batches, the discriminators' few-row GEMMs): workgroups are
            // dispatched to the 8 XCDs round-robin, so workgroup w = 8 l + x takes tile l of XCD x's CONTIGUOUS share of the tile list —
            // tiles are numbered (branch, channel group)-major, i.e. an XCD's L2 then streams an eighth of the weights instead of all
            const int x = (int)blockIdx.x & 7, l = (int)blockIdx.x >> 3;
            const int base = mp.total_tiles >> 3, rem = mp.total_tiles & 7;
            return x * base + min(x, rem) + l;
        }
        return (int)blockIdx.x + (my_rounds - 1 - i) * (int)gridDim.x;
    };
    // ragged batches: tiles past the end of their sequence are skipped by both roles (same predicate, so the barrier
    // counts stay in step); nxt(i) = first non-empty position of this workgroup's list at or after i
    auto nxt = [&](int i) {
        if (is_ragged(mp.p[0]))
            while (i < my_rounds) {
                const Tile T = decode(tile_of(i));
                if (T.t0 < seq_rows(mp.p[T.b], T.seq)) break;
                ++i;
            }
        return i;
    };
    const float* O = reinterpret_cast<const float*>(smem_b + 2 * buf_bytes);
    // Output pass of a finished tile: the MFMA waves left the raw accumulators in the LDS out-buffer as
    // O[time row][channel]; the participating threads (the 256 loader threads, or all 512 for the last tile) walk it row-major (a wave-instruction covers whole 512-byte
    // row segments), add bias and residual, and write the fp32 rows and/or the activated split rows.
    auto write_out = [&](const Tile& T, int ltid, int nthr) {
        // A CU retires roughly one vector-store wave-instruction per ~70 cycles whatever its width, so every store
        // here is 16 bytes per lane: a thread owns 8 adjacent channels of a row (2 x float4 in, 16 B hi + 16 B lo out).
        const ConvParams& p = mp.p[T.b];
        const int vc_base = T.ng * TN;                            // first virtual channel of the tile
        const int width = min(TN, p.n_blocks32 * 32 - vc_base);  // a partial channel group is narrower (32 | width)
        const int w8 = width >> 3;                                // 4, 8, 12 or 16 units per row
        const int rpp = nthr / w8;                                // rows per pass of the participating threads
        const int rr = ltid / w8;
        const int c8 = (ltid - rr * w8) * 8;                      // this thread's channels: the same in every pass
        const bool lane_on = rr < rpp;                            // w8 = 12 leaves a few threads idle
        const size_t seq_base = (size_t)T.seq * p.L;
        const float slope_out = p.slope_out;
        const int rows = min(TM, seq_rows(p, T.seq) - T.t0);
        const float* const bias_z = p.bias + (size_t)T.z * mp.zs_b;
        float* const y_z = p.y ? p.y + (size_t)T.z * mp.zs_y : nullptr;
        char* const ys_z = p.ys ? p.ys + (size_t)T.z * mp.zs_y * 4 : nullptr;
        const int vc = vc_base + c8;
        const f32x4 bv0 = *reinterpret_cast<const f32x4*>(bias_z + vc);
        const f32x4 bv1 = *reinterpret_cast<const f32x4*>(bias_z + vc + 4);
        // split rows: virtual channel -> (real row within the virtual row, channel); 8 | cout_real
        const int ph_row = vc / p.cout_real;
        const int split_off = ph_row * p.cout_real * 4 + (vc - ph_row * p.cout_real) * 2;
        constexpr int UB = 4;  // rows in flight per thread: LDS reads and residual loads are issued before any is used
        for (int r0 = 0; r0 < rows; r0 += rpp * UB) {
            f32x4 v0[UB], v1[UB], q0[UB], q1[UB], m0[UB], m1[UB];
#pragma unroll
            for (int q = 0; q < UB; ++q) {
                const int row_l = r0 + q * rpp + rr;
                v0[q] = v1[q] = q0[q] = q1[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                m0[q] = m1[q] = f32x4{1.f, 1.f, 1.f, 1.f};
                if (lane_on && row_l < rows) {
                    v0[q] = *reinterpret_cast<const f32x4*>(&O[row_l * OP + c8]);
                    v1[q] = *reinterpret_cast<const f32x4*>(&O[row_l * OP + c8 + 4]);
                    if (p.mask_src) {
                        const float* mp_ = p.mask_src + (seq_base + T.t0 + row_l) * p.cout_total + vc;
                        m0[q] = *reinterpret_cast<const f32x4*>(mp_);
                        m1[q] = *reinterpret_cast<const f32x4*>(mp_ + 4);
                    }
                    if constexpr (KS == 4) {  // (p0 + p1) + (p2 + p3)
                        const f32x4 a0 = *reinterpret_cast<const f32x4*>(&O[(TM + row_l) * OP + c8]);
                        const f32x4 a1 = *reinterpret_cast<const f32x4*>(&O[(TM + row_l) * OP + c8 + 4]);
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(&O[(2 * TM + row_l) * OP + c8]);
                        const f32x4 b1 = *reinterpret_cast<const f32x4*>(&O[(2 * TM + row_l) * OP + c8 + 4]);
                        const f32x4 d0 = *reinterpret_cast<const f32x4*>(&O[(3 * TM + row_l) * OP + c8]);
                        const f32x4 d1 = *reinterpret_cast<const f32x4*>(&O[(3 * TM + row_l) * OP + c8 + 4]);
                        v0[q] = (v0[q] + a0) + (b0 + d0);
                        v1[q] = (v1[q] + a1) + (b1 + d1);
                    }
                    if (p.res) {
                        const float* rp = p.res + (seq_base + T.t0 + row_l) * p.cout_total + vc;
                        q0[q] = *reinterpret_cast<const f32x4*>(rp);
                        q1[q] = *reinterpret_cast<const f32x4*>(rp + 4);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < UB; ++q) {
                const int row_l = r0 + q * rpp + rr;
                if (lane_on && row_l < rows) {
                    const size_t row = seq_base + T.t0 + row_l;
                    float o[8];
                    if (p.mask_src) {  // backward: act'(x) * (dgrad + 0) + skip gradient
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[e] = (v0[q][e] + bv0[e]) * (m0[q][e] > 0.f ? 1.f : p.mask_slope) + q0[q][e];
                            o[4 + e] = (v1[q][e] + bv1[e]) * (m1[q][e] > 0.f ? 1.f : p.mask_slope) + q1[q][e];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[e] = (v0[q][e] + bv0[e]) + q0[q][e];
                            o[4 + e] = (v1[q][e] + bv1[e]) + q1[q][e];
                        }
                    }
                    if (y_z) {
                        float* yp = y_z + row * p.cout_total + vc;
                        *reinterpret_cast<f32x4*>(yp) = f32x4{o[0], o[1], o[2], o[3]};
                        *reinterpret_cast<f32x4*>(yp + 4) = f32x4{o[4], o[5], o[6], o[7]};
                    }
                    if (ys_z) {
                        if constexpr (F32) {
                            float a[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) a[e] = fmaxf(o[e], o[e] * slope_out);  // LeakyReLU, 0 <= slope <= 1
                            float* orow = reinterpret_cast<float*>(ys_z) + row * (size_t)p.cout_total + vc;
                            *reinterpret_cast<f32x4*>(orow) = f32x4{a[0], a[1], a[2], a[3]};
                            *reinterpret_cast<f32x4*>(orow + 4) = f32x4{a[4], a[5], a[6], a[7]};
                        } else {
                            bf16x8 hi, lo;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float a = fmaxf(o[e], o[e] * slope_out);  // LeakyReLU, 0 <= slope <= 1
                                hi[e] = (__bf16)a;
                                lo[e] = (__bf16)(a - (float)hi[e]);
                            }
                            char* orow = ys_z + row * (size_t)p.cout_total * 4 + split_off;
                            *reinterpret_cast<bf16x8*>(orow) = hi;
                            *reinterpret_cast<bf16x8*>(orow + p.cout_real * 2) = lo;
                        }
                    }
                }
            }
        }
    };
    if (loader) {
        // ---------------- loader role: LDS-DMA of split rows + the finished tile's output pass ----------------
        const int lw = wave - NW;
        const int ltid = tid - NW * 64;

        auto dma_item = [&](const Tile& T, int c, int jj) {
            const ConvParams& p = mp.p[T.b];
            const int R = TM + p.halo;
            const int ninstr = (R * SPR + 63) >> 6;  // 1 KiB of LDS per wave-instruction
            const int Ls = p.x_rows ? p.x_rows : seq_rows(p, T.seq);
            const int row_bytes = p.x_row_bytes ? p.x_row_bytes : p.cin * 4;
            const char* const xs_z = p.xs + (size_t)T.z * mp.zs_x + (size_t)T.seq * (p.x_seq_bytes ? (size_t)p.x_seq_bytes : (size_t)p.L * p.cin * 4);
            char* dst = smem_b + (jj & 1) * buf_bytes;
            const int c0b = c * CH * 2;  // byte offset of this chunk inside the hi (and lo) half of a row
            for (int i = lw; i < ninstr; i += 4) {
                const int n = i * 64 + lane;
                const int r = n >> LOG_SPR;
                const int sl = (n & (SPR - 1)) ^ ((r >> LOG_RPB) & (SPR - 1));  // logical slot stored at this position
                const int t = T.t0 + p.off_min + r;
                const char* src = p.zeros;
                if (r < R && t >= 0 && t < Ls) {
                    const int ts = p.x_up > 1 ? (int)__umulhi((unsigned)t, p.x_up_rcp) : t;  // (nearest upsample: out[t] = in[t / s])
                    if constexpr (F32) src = xs_z + (size_t)ts * row_bytes + 2 * c0b + sl * 16;
                    else src = xs_z + (size_t)ts * row_bytes + (sl < SPR / 2 ? c0b + sl * 16 : p.cin * 2 + c0b + (sl - SPR / 2) * 16);
                }
                // aux = 2: non-temporal — an activation row is staged by one or two CUs only (+2.3 % end to end); rows that many channel groups
                // re-stage stay cacheable (MultiConvParams::stage_cached)
                if (mp.stage_cached)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
                else
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 2);
            }
        };
        // The same item from PRE-activation fp32 rows (ConvParams::act_in = n streams): 16 bytes per lane and stream through registers, the streams
        // summed in order and divided by n (n > 1: the MRF mean, hifigan.py:226-230 — the same expression as mrf_split_kernel, bit for bit),
        // LeakyReLU(slope_in) applied, written to the slot the DMA would have filled (position i * 1024 + lane * 16 holds logical slot sl of row r).
        auto stage_act_item = [&](const Tile& T, int c, int jj) {
            if constexpr (F32) {
                const ConvParams& p = mp.p[T.b];
                const int R = TM + p.halo;
                const int ninstr = (R * SPR + 63) >> 6;
                const int Ls = p.x_rows ? p.x_rows : seq_rows(p, T.seq);
                const int row_bytes = p.x_row_bytes ? p.x_row_bytes : p.cin * 4;
                const size_t seq_off = (size_t)T.z * mp.zs_x + (size_t)T.seq * (p.x_seq_bytes ? (size_t)p.x_seq_bytes : (size_t)p.L * p.cin * 4);
                char* dst = smem_b + (jj & 1) * buf_bytes;
                const int c0b = c * CH * 2;
                const float slope = p.slope_in;
                const int nsrc = p.act_in;
                auto src_off = [&](int i, bool& ok) {  // byte offset (inside a stream) of what wave-instruction i stages in this lane
                    const int n = i * 64 + lane;
                    const int r = n >> LOG_SPR;
                    const int sl = (n & (SPR - 1)) ^ ((r >> LOG_RPB) & (SPR - 1));
                    const int t = T.t0 + p.off_min + r;
                    ok = i < ninstr && r < R && t >= 0 && t < Ls;
                    const int ts = p.x_up > 1 ? (int)__umulhi((unsigned)t, p.x_up_rcp) : t;
                    return seq_off + (size_t)ts * row_bytes + 2 * c0b + sl * 16;
                };
                auto ld = [&](const char* base, size_t off) {
                    if (mp.stage_cached) return *reinterpret_cast<const f32x4*>(base + off);
                    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + off));
                };
                if (nsrc <= 1) {
                    constexpr int UB = 12;  // (an item is 36-48 wave-loads: all of a loader wave's share in flight at once, one memory latency per item like the DMA)
                    for (int i0 = lw; i0 < ninstr; i0 += 4 * UB) {
                        f32x4 v[UB];
#pragma unroll
                        for (int q = 0; q < UB; ++q) {
                            bool ok;
                            const size_t off = src_off(i0 + 4 * q, ok);
                            v[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (ok) v[q] = ld(p.xs, off);
                        }
#pragma unroll
                        for (int q = 0; q < UB; ++q) {
                            const int i = i0 + 4 * q;
                            if (i < ninstr) {
                                f32x4 a;
#pragma unroll
                                for (int e = 0; e < 4; ++e) a[e] = fmaxf(v[q][e], v[q][e] * slope);
                                *reinterpret_cast<f32x4*>(dst + i * 1024 + lane * 16) = a;
                            }
                        }
                    }
                } else {
                    constexpr int UB = 4;  // x up to four streams in flight
                    for (int i0 = lw; i0 < ninstr; i0 += 4 * UB) {
                        f32x4 v[UB], w1[UB], w2[UB], w3[UB];
#pragma unroll
                        for (int q = 0; q < UB; ++q) {
                            bool ok;
                            const size_t off = src_off(i0 + 4 * q, ok);
                            v[q] = w1[q] = w2[q] = w3[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (ok) {
                                v[q] = ld(p.xs, off);
                                w1[q] = ld(p.xs_more[0], off);
                                if (nsrc >= 3) w2[q] = ld(p.xs_more[1], off);
                                if (nsrc == 4) w3[q] = ld(p.xs_more[2], off);
                            }
                        }
#pragma unroll
                        for (int q = 0; q < UB; ++q) {
                            const int i = i0 + 4 * q;
                            if (i < ninstr) {
                                f32x4 a;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    float m;
                                    if (nsrc == 4) m = (((v[q][e] + w1[q][e]) + w2[q][e]) + w3[q][e]) / 4.0f;
                                    else if (nsrc == 3) m = ((v[q][e] + w1[q][e]) + w2[q][e]) / 3.0f;
                                    else m = (v[q][e] + w1[q][e]) / 2.0f;
                                    a[e] = fmaxf(m, m * slope);
                                }
                                *reinterpret_cast<f32x4*>(dst + i * 1024 + lane * 16) = a;
                            }
                        }
                    }
                }
            }
        };
        auto stage_item = [&](const Tile& T, int c, int jj) {
            if constexpr (F32) {
                if (mp.p[T.b].act_in) {
                    stage_act_item(T, c, jj);
                    return;
                }
            }
            dma_item(T, c, jj);
        };
        // items are numbered j = 0.. over (tile, chunk); the DMA of item j+1 runs while the MFMA waves compute item j
        int j = 0;
        HIFICAR_STAMP(0);
        bool have_prev = false;
        Tile Tprev;
        for (int it = nxt(0); it < my_rounds; it = nxt(it + 1)) {
            const Tile T = decode(tile_of(it));
            for (int c = 0; c < nchunks; ++c, ++j) {
                stage_item(T, c, j);
                // the tile that finished with item j-2 published its accumulators at barrier j-1 (nchunks >= 2, so the
                // MFMA waves cannot overwrite the out-buffer before barrier j+1)
                if constexpr (!DOUT) {
                    if (c == 1 && have_prev) write_out(Tprev, ltid, 256);
                }
                HIFICAR_STAMP(1 + 2 * j);
                __syncthreads();  // item j landed (hipcc drains vmcnt before the barrier); MFMA waves are done with item j-1's buffer
                HIFICAR_STAMP(2 + 2 * j);
            }
            Tprev = T;
            have_prev = true;
        }
        __syncthreads();  // the last tile's accumulators are in the out-buffer
        if constexpr (!DOUT) {
            if (have_prev) write_out(Tprev, tid, NTHR);  // all waves share the final output pass (nothing left to hide it behind)
        }
        HIFICAR_STAMP(63);
        return;
    }

    // ---------------- MFMA role ----------------
    __builtin_amdgcn_s_setprio(1);  // the MFMA wave outranks the loader wave sharing its SIMD for issue slots
    f32x16 acc[NB][MI];
    // weight fragments: wr[q][u] holds slab u's fragments (of the wave's channel block q) for its NEXT use; right after a slab's MFMAs have been issued its registers are
    // reloaded for the following tap (one tap = NC16 slab steps ahead).  One register set per slab and no rotation: a two-deep ring
    // needs register copies at every loop back-edge, and hipcc waits for the just-issued loads there (a full L2 latency per item).
    using frag_t = typename std::conditional<F32, f32x4, bf16x8>::type;  // 16 bytes per lane either way
    constexpr int NMF = (F32 ? 8 : 3) * MI * NB;  // MFMAs of one K-slab step
    frag_t wr[NB][NC16][2];
    auto wstream = [&](const Tile& T) {
        const ConvParams& p = mp.p[T.b];
        const int nb = (T.ng * WN + wn) * NB;
        return reinterpret_cast<const frag_t*>(reinterpret_cast<const char*>(p.w16) + (size_t)T.z * mp.zs_w) + (size_t)(nb < p.n_blocks32 ? nb : 0) * p.ntaps * (p.cin / 16) * 128 + lane;
    };
    // NB = 2: the second channel block's stream lies one block's worth of fragments behind the first's (0 when the layer has no such block:
    // a partial channel group — the wave then multiplies the first block twice and the output pass ignores the columns)
    auto wstride = [&](const Tile& T) {
        const ConvParams& p = mp.p[T.b];
        return ((T.ng * WN + wn) * NB + 1 < p.n_blocks32) ? (long long)p.ntaps * (p.cin / 16) * 128 : 0LL;
    };
    const frag_t* wp = nullptr;   // where the next tap-group of weight fragments is loaded from
    const frag_t* wp2 = nullptr;  // ... of the second channel block (NB = 2)
    int groups_left = 0;          // tap-groups of the current tile's stream not yet requested
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

    // LDS byte addresses of this lane's activation fragments for one tap: [K slab][hi|lo]
    auto addr_set = [&](int buf_off, int roff, int (&ad)[NC16][2]) {
        const int r0 = wave_row0 + li + roff;
        const int swz = (r0 >> LOG_RPB) & (SPR - 1);
        const int base = buf_off + r0 * RB;
#pragma unroll
        for (int u = 0; u < NC16; ++u) {
            if constexpr (F32) {  // slab u = slots 4u .. 4u+3: half v, half-wave g
                ad[u][0] = base + (((4 * u + g) ^ swz) << 4);
                ad[u][1] = base + (((4 * u + 2 + g) ^ swz) << 4);
            } else {              // [hi | lo] halves of the row
                ad[u][0] = base + (((2 * u + g) ^ swz) << 4);
                ad[u][1] = base + (((SPR / 2 + 2 * u + g) ^ swz) << 4);
            }
        }
    };
    auto load_x = [&](frag_t (&xh)[MI], frag_t (&xl)[MI], const int (&ad)[2]) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {  // +32 rows keeps the swizzle (32 % 16 == 0): immediate offsets
            xh[mi] = *reinterpret_cast<const frag_t*>(smem_b + ad[0] + mi * 32 * RB);
            xl[mi] = *reinterpret_cast<const frag_t*>(smem_b + ad[1] + mi * 32 * RB);
        }
    };
    auto mfma_step = [&](const frag_t (&xh)[MI], const frag_t (&xl)[MI], const frag_t (&wh)[NB], const frag_t (&wl)[NB]) {
        // consecutive MFMAs never share an accumulator (MI * NB > 1)
        if constexpr (F32) {
            // (xh, wh) = channels 0..7 of the slab, (xl, wl) = channels 8..15; step s multiplies channels s and 4 + s
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int q = 0; q < NB; ++q)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        // direct output (kRowMajorAcc): activations as the A operand, so that a lane ends up with ONE channel and 16 time rows per
                        // block — the same products summed in the same order, D = X W^T instead of D^T = W X^T
                        if constexpr (kRowMajorAcc) acc[q][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(xh[mi][s4], wh[q][s4], acc[q][mi], 0, 0, 0);
                        else acc[q][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(wh[q][s4], xh[mi][s4], acc[q][mi], 0, 0, 0);
                    }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int q = 0; q < NB; ++q)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        if constexpr (kRowMajorAcc) acc[q][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(xl[mi][s4], wl[q][s4], acc[q][mi], 0, 0, 0);
                        else acc[q][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[q][s4], xl[mi][s4], acc[q][mi], 0, 0, 0);
                    }
        } else {
#pragma unroll
            for (int q = 0; q < NB; ++q)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[q][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[q], xl[mi], acc[q][mi], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NB; ++q)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[q][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[q], xh[mi], acc[q][mi], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NB; ++q)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[q][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[q], xh[mi], acc[q][mi], 0, 0, 0);
        }
    };
    // one K-slab step of the ring: take slab u's fragments, reload the registers for the following tap, multiply
    auto slab_step = [&](const frag_t (&xh)[MI], const frag_t (&xl)[MI], int u) {
        frag_t wh[NB], wl[NB];
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
        pin_slab_step<2 * MI, 2 * NB, NMF>();
    };

    // Direct output with transposed accumulators (kRowMajorAcc): the epilogue of one tile, behind its K loop.  (Issuing it between the MFMAs of the
    // tile's last tap was built and measured slower, profiles/r05_epilogue_layouts.txt; tools/r05_kernels/ keeps that form.)
    auto epilogue_rm = [&](const Tile& T, const ConvParams& p, int nb, int rows_valid, float bias_l) {
        if constexpr (kRowMajorAcc) {
            // a row block of the wave's share of the tile as raw buffers over its VALID rows: accesses to rows past a sequence's end fall outside the
            // range (loads return 0, stores are dropped) — no per-row branches.  Descriptors are built from wave-uniform scalars where they are used
            // (carried across a loop they end up in vector registers and every access in a readfirstlane loop).
            const int rows_w = __builtin_amdgcn_readfirstlane(max(min(rows_valid - wave_row0, MI * 32), 0));
            const unsigned pitch_b = __builtin_amdgcn_readfirstlane((unsigned)p.cout_total * 4u);
            const size_t first = ((size_t)T.seq * p.L + T.t0 + wave_row0) * p.cout_total;
            float* const y_p = p.y;
            char* const ys_p = p.ys;
            const float* const res_p = p.res;
            auto uni64 = [](unsigned long long v) {
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
                return ((unsigned long long)hi << 32) | lo;
            };
            const bool has_y = y_p != nullptr, has_ys = ys_p != nullptr, has_res = res_p != nullptr, has_mask = p.mask_src != nullptr;
            const unsigned long long b_y = uni64((unsigned long long)(has_y ? y_p + (size_t)T.z * mp.zs_y + first : nullptr));
            const unsigned long long b_ys = uni64((unsigned long long)(has_ys ? reinterpret_cast<float*>(ys_p) + (size_t)T.z * mp.zs_y + first : nullptr));
            const unsigned long long b_res = uni64((unsigned long long)(has_res ? res_p + first : nullptr));
            const unsigned long long b_mask = uni64((unsigned long long)(has_mask ? p.mask_src + first : nullptr));
            auto rsrc_of = [&](unsigned long long base, bool present, int mi) {  // row block mi of an operand
                const unsigned rows = (unsigned)max(min(rows_w - mi * 32, 32), 0);
                return __builtin_amdgcn_make_buffer_rsrc((void*)uni64(base + (unsigned long long)(mi * 32) * pitch_b), 0,
                                                         __builtin_amdgcn_readfirstlane(present ? rows * pitch_b : 0u), 0x00020000);
            };
            constexpr int kAux = 0;
            const int voff = (int)((4u * g * p.cout_total + nb * 32 + li) * 4u);  // (row 4 g, this lane's channel)
            int off16[16];  // byte offset of register r's element inside a row block: row 8 (r >> 2) + (r & 3) + 4 g
#pragma unroll
            for (int r = 0; r < 16; ++r) off16[r] = voff + (int)((8 * (r >> 2) + (r & 3)) * pitch_b);
            const float slope_out = p.slope_out, mask_slope = p.mask_slope;
            // one straight-line pass per combination of operands (HR residual, HM mask, HY fp32 rows, HS activated rows): a pass compiled for all
            // four with per-element uniform branches is slower than the 16-byte form it replaces.  An operand a pass was compiled with but the
            // layer lacks has an empty range.
            auto pass = [&](auto HR, auto HM, auto HY, auto HS) {
                constexpr bool kRes = decltype(HR)::value, kMask = decltype(HM)::value, kY = decltype(HY)::value, kYs = decltype(HS)::value;
                auto fetch1 = [&](int mi, int r, float (&rs)[16], float (&mk)[16]) {  // residual / mask value of element r of row block mi
                    if constexpr (kRes) rs[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_of(b_res, has_res, mi), off16[r], 0, 0));
                    if constexpr (kMask) mk[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_of(b_mask, has_mask, mi), off16[r], 0, 0));
                };
                auto finish = [&](int mi, int r, const float (&rs)[16], const float (&mk)[16]) {  // element r of row block mi
                    float o = acc[0][mi][r] + bias_l;
                    if constexpr (kMask) o *= mk[r] > 0.f ? 1.f : mask_slope;  // backward: act'(x) * dgrad + skip gradient
                    if constexpr (kRes) o += rs[r];
                    if constexpr (kY) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rsrc_of(b_y, has_y, mi), off16[r], 0, kAux);
                    // LeakyReLU as v_mul + v_med3 (max(o, slope o) = the median of o, slope o and +inf; fmaxf() canonicalises o first: a third instruction)
                    if constexpr (kYs) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, __builtin_amdgcn_fmed3f(o, o * slope_out, __builtin_inff())), rsrc_of(b_ys, has_ys, mi), off16[r], 0, kAux);
                };
                {
                    constexpr int G2 = MI >= 2 ? 2 : 1;
#pragma unroll
                    for (int m0 = 0; m0 < MI; m0 += G2) {
                        float rs[G2][16], mk[G2][16];
                        if constexpr (kRes || kMask) {
#pragma unroll
                            for (int mm = 0; mm < G2; ++mm)
#pragma unroll
                                for (int r = 0; r < 16; ++r) fetch1(m0 + mm, r, rs[mm], mk[mm]);
                        }
#pragma unroll
                        for (int mm = 0; mm < G2; ++mm)
#pragma unroll
                            for (int r = 0; r < 16; ++r) finish(m0 + mm, r, rs[mm], mk[mm]);
                    }
                }
            };
            constexpr std::true_type Y{};
            constexpr std::false_type N{};
            if (has_mask) {  // (a data-gradient launch)
                pass(Y, Y, Y, Y);
                return;
            }
            if (has_res) {
                if (has_y && has_ys) pass(Y, N, Y, Y);
                else if (has_y) pass(Y, N, Y, N);
                else pass(Y, N, N, Y);
            } else {
                if (has_y && has_ys) pass(N, N, Y, Y);
                else if (has_y) pass(N, N, Y, N);
                else pass(N, N, N, Y);
            }
        }
    };
    auto bias_of = [&](const Tile& T) {
        const int nb = (T.ng * WN + wn) * NB;
        return (mp.p[T.b].bias + (size_t)T.z * mp.zs_b)[(nb < mp.p[T.b].n_blocks32 ? nb : 0) * 32 + li];
    };
    int j = 0;
    int last = -1;  // position of the last tile computed
    HIFICAR_STAMP(0);
    if constexpr (KS == 4) {
        // ---------------- split-K: wave ks computes the steps s = ks, ks + 4, ... of every item ----------------
        const int ks = cw;
        constexpr int LOG_NC = NC16 == 4 ? 2 : NC16 == 2 ? 1 : 0;
        for (int it = nxt(0); it < my_rounds; it = nxt(it + 1)) {
            last = it;
            const Tile T = decode(tile_of(it));
            const ConvParams& p = mp.p[T.b];
            const int nb = T.ng;  // one 32-channel block per tile
            const int phase = (int)((unsigned)nb / (unsigned)p.nb32_per_phase);
            const int roff0 = __builtin_amdgcn_readfirstlane(p.tap_off0[phase] - p.off_min);
            const int tap_step = p.tap_step;
            const int nsteps = p.ntaps * NC16;
            const frag_t* wtile = reinterpret_cast<const frag_t*>(reinterpret_cast<const char*>(p.w16) + (size_t)T.z * mp.zs_w) + (size_t)nb * nchunks * nsteps * 128 + lane;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][mi][r] = 0.f;
            auto x_addr = [&](int buf_off, int s, int (&ad)[2]) {
                const int t = s >> LOG_NC, u = s & (NC16 - 1);
                const int r0 = li + roff0 + t * tap_step;
                const int swz = (r0 >> LOG_RPB) & (SPR - 1);
                const int base = buf_off + r0 * RB;
                if constexpr (F32) {
                    ad[0] = base + (((4 * u + g) ^ swz) << 4);
                    ad[1] = base + (((4 * u + 2 + g) ^ swz) << 4);
                } else {
                    ad[0] = base + (((2 * u + g) ^ swz) << 4);
                    ad[1] = base + (((SPR / 2 + 2 * u + g) ^ swz) << 4);
                }
            };
            const int spi = ks < nsteps ? (nsteps - ks + 3) >> 2 : 0;  // this wave's steps per item: s = ks + 4 i
            for (int c = 0; c < nchunks; ++c, ++j) {
                // The weight fragments do not depend on the staging barrier: this item's first two steps are requested before it.
                // Inside the item they run two steps ahead of their use in two alternating register sets (no rotation copies).
                const frag_t* wc = wtile + (size_t)c * nsteps * 128 + (size_t)ks * 128;  // step i of this wave: wc[i * 512 (+ 64)]
                frag_t w0h, w0l, w1h, w1l;
                if (spi > 0) {
                    w0h = wc[0];
                    w0l = wc[64];
                }
                if (spi > 1) {
                    w1h = wc[512];
                    w1l = wc[512 + 64];
                }
                HIFICAR_STAMP(1 + 3 * j);
                // item j is staged (the weight fragments requested just above stay in flight across it: __syncthreads() would wait for them here)
                lds_barrier();
                HIFICAR_STAMP(2 + 3 * j);
                const int buf_off = (j & 1) * buf_bytes;
                frag_t x0h[MI], x0l[MI], x1h[MI], x1l[MI];
                int ad[2];
                if (spi > 0) {
                    x_addr(buf_off, ks, ad);
                    load_x(x0h, x0l, ad);
                }
                for (int i = 0; i < spi; i += 2) {
                    if (i + 1 < spi) {
                        x_addr(buf_off, ks + 4 * (i + 1), ad);
                        load_x(x1h, x1l, ad);
                    }
                    {
                        const frag_t cwh[1] = {w0h}, cwl[1] = {w0l};
                        if (i + 2 < spi) {
                            w0h = wc[(size_t)(i + 2) * 512];
                            w0l = wc[(size_t)(i + 2) * 512 + 64];
                        }
                        mfma_step(x0h, x0l, cwh, cwl);
                    }
                    if (i + 1 < spi) {
                        if (i + 2 < spi) {
                            x_addr(buf_off, ks + 4 * (i + 2), ad);
                            load_x(x0h, x0l, ad);
                        }
                        const frag_t cwh[1] = {w1h}, cwl[1] = {w1l};
                        if (i + 3 < spi) {
                            w1h = wc[(size_t)(i + 3) * 512];
                            w1l = wc[(size_t)(i + 3) * 512 + 64];
                        }
                        mfma_step(x1h, x1l, cwh, cwl);
                    }
                }
            }
            {   // this wave's partial sums -> out-buffer slice ks
                float* Ow = reinterpret_cast<float*>(smem_b + 2 * buf_bytes) + ks * TM * OP;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[0][mi][4 * q + e];
                        *reinterpret_cast<f32x4*>(&Ow[(mi * 32 + li) * OP + 8 * q + 4 * g]) = v;
                    }
            }
        }
        HIFICAR_STAMP(3 * j);
        __syncthreads();  // matches the loader waves' final barrier
        HIFICAR_STAMP(62);
        if (last >= 0) write_out(decode(tile_of(last)), tid, NTHR);
        HIFICAR_STAMP(63);
        return;
    }
    Tile T_carry = {};  // the tile about to be computed: decoded once, as the previous tile's `Tn`
    // The weight ring is primed ONCE, for the first tile; every tile's last tap then loads the next tile's head.  A wave without a channel block of its
    // own in a tile (partial channel group) runs the K loop all the same — on block 0's weights, results dropped — so that the ring never has to be
    // primed inside the tile loop: a second definition of the ring registers there makes hipcc copy all 32 of them at every tile's start, behind a
    // vmcnt(0) that also sits out the previous tile's stores.
    // (direct-output kernels only — kPrimeOnce: in the other forms — out-buffer, register-blocked, bf16x3 — the MFMA waves store nothing, partial channel
    // groups are common (C = 64 on a 128-channel tile) and an idle wave's duplicate LDS reads cost more than the copies: bf16x3 leg 100 -> 93 M samples/s)
    bool primed = false;  // (!kPrimeOnce) the ring holds the head of the tile about to be computed
    (void)primed;
    if (nxt(0) < my_rounds) {
        T_carry = decode(tile_of(nxt(0)));
        if constexpr (kPrimeOnce) prime(T_carry);
    }
    for (int it = nxt(0), itn; it < my_rounds; it = itn) {
        itn = nxt(it + 1);
        last = it;
        const Tile T = T_carry;
        const ConvParams& p = mp.p[T.b];
        const int nb = (T.ng * WN + wn) * NB;
        const bool active = nb < p.n_blocks32;
        const int phase = active ? (int)((unsigned)nb / (unsigned)p.nb32_per_phase) : 0;
        const int roff0 = __builtin_amdgcn_readfirstlane(p.tap_off0[phase] - p.off_min);
        const int tap_step = p.tap_step;
        const int ntaps = p.ntaps;
        // (read here, not in the epilogue: a ragged batch's length is a memory load)
        const int rows_valid_t = kRowMajorAcc ? __builtin_amdgcn_readfirstlane(min(TM, seq_rows(p, T.seq) - T.t0)) : 0;
        // after this tile's stream is exhausted the loads continue with the NEXT tile's first tap-groups, so its ring is
        // primed when it starts (the last tile re-reads its own head: harmless)
        const Tile Tn = itn < my_rounds ? decode(tile_of(itn)) : T;
        T_carry = Tn;
        const frag_t* wp_next = wstream(Tn);
        const long long wst_next = NB == 2 ? wstride(Tn) : 0LL;
        const int groups_next = nchunks * mp.p[Tn.b].ntaps;
        if constexpr (!kPrimeOnce) {
            if (active && !primed) prime(T);  // first tile, or this wave sat out the previous tile (partial channel group)
        }
        // Direct output with transposed accumulators: a lane owns ONE channel, so the accumulators start at its bias (one value per lane for all MI x 16
        // registers) and the epilogue has no bias to add or wait for; the NEXT tile's value is requested now and arrives behind this tile's K loop.
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][mi][r] = 0.f;
        // (kRowMajorAcc) this lane's bias, requested here and first used in the epilogue: the load has the whole K loop to arrive.  (Requested a tile
        // ahead and folded into the accumulators' start value it costs more than it saves: hipcc waits for the loop-carried load with a vmcnt that
        // also sits out every store of the previous tile's epilogue.)  A wave without a channel block of its own reads block 0's and never uses it.
        float bias_l = 0.f;
        if constexpr (kRowMajorAcc) bias_l = bias_of(T);

        for (int c = 0; c < nchunks; ++c, ++j) {
            HIFICAR_STAMP(1 + 3 * j);
            // item j is staged.  Direct output: the barrier alone — __syncthreads() also waits for this wave's own global stores (vmcnt(0): the whole
            // write latency of the tile just stored, with the matrix pipe idle); they only have to land by the end of the kernel, and what the barrier
            // orders here is LDS (the loaders wait for their DMA in front of theirs).
            lds_barrier();
            HIFICAR_STAMP(2 + 3 * j);
            if constexpr (!kPrimeOnce) {
                if (!active) continue;  // partial channel group: this wave only keeps the barriers
            }
            const int buf_off = (j & 1) * buf_bytes;
            int ad[NC16][2];
            addr_set(buf_off, roff0, ad);
            if constexpr (NC16 % 2 == 0) {
                frag_t x0h[MI], x0l[MI], x1h[MI], x1l[MI];
                load_x(x0h, x0l, ad[0]);
                for (int t = 0; t < ntaps; ++t) {
                    const bool last_tap = t + 1 == ntaps;
                    int adn[NC16][2];
                    addr_set(buf_off, roff0 + (last_tap ? t : t + 1) * tap_step, adn);
                    if (groups_left == 0) {  // stream exhausted: continue with the next tile's head
                        wp = wp_next;
                        if constexpr (NB == 2) wp2 = wp_next + wst_next;
                        groups_left = groups_next;
                    }
                    --groups_left;
#pragma unroll
                    for (int u = 0; u < NC16; u += 2) {
                        load_x(x1h, x1l, ad[u + 1]);
                        slab_step(x0h, x0l, u);
                        if (u + 2 < NC16) load_x(x0h, x0l, ad[u + 2]);
                        else load_x(x0h, x0l, adn[0]);
                        slab_step(x1h, x1l, u + 1);
                    }
                    wp += NC16 * 128;
                    if constexpr (NB == 2) wp2 += NC16 * 128;
#pragma unroll
                    for (int u = 0; u < NC16; ++u) {
                        ad[u][0] = adn[u][0];
                        ad[u][1] = adn[u][1];
                    }
                }
            } else {
                // one K slab per tap (NC16 == 1): the fragments of tap t+1 are read while tap t's MFMAs issue
                static_assert(NC16 % 2 == 0 || NC16 == 1, "odd chunk sizes other than 16 channels are not instantiated");
                frag_t x0h[MI], x0l[MI], x1h[MI], x1l[MI];
                load_x(x0h, x0l, ad[0]);
                auto tap = [&](const frag_t (&xh)[MI], const frag_t (&xl)[MI]) {
                    if (groups_left == 0) {
                        wp = wp_next;
                        if constexpr (NB == 2) wp2 = wp_next + wst_next;
                        groups_left = groups_next;
                    }
                    --groups_left;
                    slab_step(xh, xl, 0);
                    wp += NC16 * 128;
                    if constexpr (NB == 2) wp2 += NC16 * 128;
                };
                for (int t = 0; t < ntaps; t += 2) {
                    if (t + 1 < ntaps) {
                        addr_set(buf_off, roff0 + (t + 1) * tap_step, ad);
                        load_x(x1h, x1l, ad[0]);
                    }
                    tap(x0h, x0l);
                    if (t + 1 < ntaps) {
                        if (t + 2 < ntaps) {
                            addr_set(buf_off, roff0 + (t + 2) * tap_step, ad);
                            load_x(x0h, x0l, ad[0]);
                        }
                        tap(x1h, x1l);
                    }
                }
            }
        }
        HIFICAR_STAMP(3 * j);
        primed = active;  // (!kPrimeOnce) an active tile ends with the ring holding the next tile's head
        if constexpr (kRowMajorAcc) {
            if (active) epilogue_rm(T, p, nb, rows_valid_t, bias_l);
        } else if (active) {
            // hand the raw accumulators to the loader waves through the LDS out-buffer O[time row][channel]:
            // lane (li, g) holds time column li and channels 8q + 4g + {0..3} in acc[mi][4q..4q+3]
            float* O = reinterpret_cast<float*>(smem_b + 2 * buf_bytes);
#pragma unroll
            for (int b2 = 0; b2 < NB; ++b2)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[b2][mi][4 * q + e];
                        *reinterpret_cast<f32x4*>(&O[(wave_row0 + mi * 32 + li) * OP + (wn * NB + b2) * 32 + 8 * q + 4 * g]) = v;
                    }
        }
    }
    lds_barrier();  // matches the loader waves' final barrier
    HIFICAR_STAMP(62);
    if constexpr (!DOUT) {
        if (last >= 0) write_out(decode(tile_of(last)), tid, NTHR);
    }
    HIFICAR_STAMP(63);
}

// direct-output form (DOUT): the dense exact-fp32 launches
template <int MI, int WM, int WN, int NC16>
__global__ __launch_bounds__((WM * WN + 4) * 64) void conv_f32do_kernel(const MultiConvParams mp) {
    conv_ws_body<MI, WM, WN, NC16, true, 1, 1, true>(mp);
}

template <int MI, int WM, int WN, int NC16>
__global__ __launch_bounds__((WM * WN + 4) * 64) void conv_bf16x3_kernel(const MultiConvParams mp) {
    conv_ws_body<MI, WM, WN, NC16, false>(mp);
}

// register-blocked wave tiles (NB = 2): tile = WM*MI*32 rows x WN*64 channels
template <int MI, int WM, int WN, int NC16>
__global__ __launch_bounds__((WM * WN + 4) * 64) void conv_bf16x3nb_kernel(const MultiConvParams mp) {
    conv_ws_body<MI, WM, WN, NC16, false, 1, 2>(mp);
}

// split-K forms (small launches): tile = MI*32 rows x 32 channels, the four MFMA waves split the K loop
template <int MI, int NC16>
__global__ __launch_bounds__(512) void conv_sk_bf16x3_kernel(const MultiConvParams mp) {
    conv_ws_body<MI, 1, 1, NC16, false, 4>(mp);
}

template <int MI, int NC16>
__global__ __launch_bounds__(512) void conv_sk_f32_kernel(const MultiConvParams mp) {
    conv_ws_body<MI, 1, 1, NC16, true, 4>(mp);
}

// ------------------------------------------------------------------------------------------------
// Fused ResBlock layer pair for narrow stages (C = 32 or 64 channels):
//     x_new = x + conv2(LeakyReLU(conv1(LeakyReLU(x))))            (residual_block.py:217-221, one dilation)
// in ONE persistent kernel: the intermediate activation never leaves the CU.  At C <= 64 the layer-by-layer
// kernels are bound by the CU's memory path (the loader waves), not by the matrix pipe; fusing the pair removes
// the intermediate's round trip (-1/3 of the bytes) and halves the launches of those stages.
//
// Same roles as conv_bf16x3_kernel.  A tile is TMc = WM*MI*32 rows of conv1 output for ALL C channels (TN = C, one
// K chunk = C): conv1 accumulates from the DMA-staged input buffer, its epilogue applies bias + LeakyReLU + hi/lo
// split and writes the LDS intermediate TS (zero rows outside the sequence = conv2's padding); conv2 then runs
// its k taps over TS and drops raw accumulators for the TMc - (k-1) valid rows into the out-buffer.
// The out-buffer ALIASES the intermediate (LDS is what limits the tile height, and a taller tile halves the weight
// traffic per MFMA): four barriers per tile —
//   A  input landed                      | loaders: write_out(previous tile) out of the shared region, hidden behind conv1
//   F  shared region free (write_out done) -> conv1's epilogue writes TS
//   B  TS complete, input buffer free    | loaders: DMA(next tile), hidden behind conv2
//   C  every wave done reading TS        -> conv2's accumulators overwrite the region as the out-buffer.
// ------------------------------------------------------------------------------------------------
struct PairParams {
    ConvParams p1[3];      // conv1 of each branch: xs, w16, bias, L, cin (= C), ntaps, tap_step, tap_off0[0], off_min, halo, zeros
    ConvParams p2[3];      // conv2: w16, bias, ntaps (= k), res, y, ys, slope_out, cout_total = cout_real = C
    int n_branches;
    int nseq;
    int tile_start[4];     // first tile id of each branch; [n_branches] = total
    int tiles_per_seq[3];  // ceil(L / (TMc - (k_b - 1)))
    int in_bytes;          // LDS bytes of the input buffer (sized for the widest halo)
    int ts_bytes;          // LDS bytes of the region shared by the intermediate and the out-buffer
    float slope_mid;       // LeakyReLU slope between conv1 and conv2
    const int* sched_start;  // host-computed schedule, as in MultiConvParams
    const int* sched_tiles;
    unsigned long long* trace;
};

// (Round 4 measured the direct-output epilogue of conv_ws_body here as well — conv2's result stored straight from the accumulators, barriers F and C
// and the loaders' output pass gone: 119.2-120.0 us per C = 32 launch against 112.2-113.0 us.  The pair tile's K loops are short (C = 32: 2 slabs
// per tap) and the matrix pipe idles through the epilogue's loads and stores, which the loader waves otherwise hide behind conv1.  Not kept.)
template <int MI, int WM, int WN, int NC16, bool F32>
__device__ __forceinline__ void conv_pair_body(const PairParams& mp) {
    static_assert(WM * WN == 4, "4 MFMA waves per workgroup");
    static_assert(NC16 == 2 || NC16 == 4, "C = 32 or 64");
    static_assert(WN * 32 == NC16 * 16, "the workgroup covers all C channels");
    constexpr int TMc = WM * MI * 32;
    constexpr int kFirstLoader = 4;
    (void)kFirstLoader;
    constexpr int CH = NC16 * 16;
    constexpr int RB = CH * 4;
    constexpr int SPR = CH / 4;
    constexpr int LOG_SPR = NC16 == 4 ? 4 : 3;
    constexpr int LOG_RPB = 4 - LOG_SPR;
    constexpr int TN = CH;
    constexpr int OP = TN + 4;
    extern __shared__ __attribute__((aligned(1024))) char smem_b[];
    const int in_bytes = mp.in_bytes;
    const int ts_off = in_bytes;
    const int o_off = in_bytes;  // the out-buffer aliases the intermediate (see the barrier protocol above)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int cw = wave & 3;
    const int wm = cw / WN;
    const int wn = cw % WN;
    const int li = lane & 31;
    const int g = lane >> 5;
    const int total_tiles = mp.tile_start[mp.n_branches];

    struct Tile {
        int b, seq, t0, tmo;
    };
    auto decode = [&](int tile) {
        Tile T;
        T.b = 0;
        if (mp.n_branches > 1 && tile >= mp.tile_start[1]) T.b = 1;
        if (mp.n_branches > 2 && tile >= mp.tile_start[2]) T.b = 2;
        const unsigned m = (unsigned)(tile - mp.tile_start[T.b]);
        const unsigned tps = (unsigned)mp.tiles_per_seq[T.b];
        T.tmo = TMc - (mp.p2[T.b].ntaps - 1);
        const unsigned seq = m / tps;  // (unsigned: half the scalar instructions of a signed division)
        T.seq = (int)seq;
        T.t0 = (int)(m - seq * tps) * T.tmo;
        return T;
    };
    // tile walk: host schedule or round-robin, light first (see conv_bf16x3_kernel)
    int sched_lo = 0, sched_hi = 0;
    if (mp.sched_start) scalar_load_2i32(mp.sched_start + blockIdx.x, sched_lo, sched_hi);
    const int my_rounds = mp.sched_start ? sched_hi - sched_lo
                                         : (total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto tile_of = [&](int it) {
        int i = it;
        if ((blockIdx.x & 1) && my_rounds >= 3 && it >= my_rounds - 2) i = it == my_rounds - 1 ? my_rounds - 2 : my_rounds - 1;
        if (mp.sched_start) return scalar_load_i32(mp.sched_tiles + sched_lo + i);
        return (int)blockIdx.x + (my_rounds - 1 - i) * (int)gridDim.x;
    };
    auto nxt = [&](int i) {  // first non-empty position at or after i (ragged batches; see conv_ws_body)
        if (is_ragged(mp.p1[0]))
            while (i < my_rounds) {
                const Tile T = decode(tile_of(i));
                if (T.t0 < seq_rows(mp.p1[T.b], T.seq)) break;
                ++i;
            }
        return i;
    };
    const int first = nxt(0);
    const float* O = reinterpret_cast<const float*>(smem_b + o_off);
    auto write_out = [&](const Tile& T, int ltid, int nthr) {
        const ConvParams& p = mp.p2[T.b];
        constexpr int w8 = TN / 8;
        const int rpp = nthr / w8;
        const int rr = ltid / w8;
        const int c8 = (ltid - rr * w8) * 8;
        const size_t seq_base = (size_t)T.seq * p.L;
        const float slope_out = p.slope_out;
        const int rows = min(T.tmo, seq_rows(p, T.seq) - T.t0);
        const f32x4 bv0 = *reinterpret_cast<const f32x4*>(p.bias + c8);
        const f32x4 bv1 = *reinterpret_cast<const f32x4*>(p.bias + c8 + 4);
        constexpr int UB = 4;
        for (int r0 = 0; r0 < rows; r0 += rpp * UB) {
            f32x4 v0[UB], v1[UB], q0[UB], q1[UB];
#pragma unroll
            for (int q = 0; q < UB; ++q) {
                const int row_l = r0 + q * rpp + rr;
                v0[q] = v1[q] = q0[q] = q1[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (row_l < rows) {
                    v0[q] = *reinterpret_cast<const f32x4*>(&O[row_l * OP + c8]);
                    v1[q] = *reinterpret_cast<const f32x4*>(&O[row_l * OP + c8 + 4]);
                    const float* rp = p.res + (seq_base + T.t0 + row_l) * TN + c8;
                    q0[q] = *reinterpret_cast<const f32x4*>(rp);
                    q1[q] = *reinterpret_cast<const f32x4*>(rp + 4);
                }
            }
#pragma unroll
            for (int q = 0; q < UB; ++q) {
                const int row_l = r0 + q * rpp + rr;
                if (row_l < rows) {
                    const size_t row = seq_base + T.t0 + row_l;
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = (v0[q][e] + bv0[e]) + q0[q][e];
                        o[4 + e] = (v1[q][e] + bv1[e]) + q1[q][e];
                    }
                    float* yp = p.y + row * TN + c8;
                    *reinterpret_cast<f32x4*>(yp) = f32x4{o[0], o[1], o[2], o[3]};
                    *reinterpret_cast<f32x4*>(yp + 4) = f32x4{o[4], o[5], o[6], o[7]};
                    if (p.ys) {
                        if constexpr (F32) {
                            float a[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) a[e] = fmaxf(o[e], o[e] * slope_out);
                            float* orow = reinterpret_cast<float*>(p.ys) + row * (size_t)TN + c8;
                            *reinterpret_cast<f32x4*>(orow) = f32x4{a[0], a[1], a[2], a[3]};
                            *reinterpret_cast<f32x4*>(orow + 4) = f32x4{a[4], a[5], a[6], a[7]};
                        } else {
                            bf16x8 hi, lo;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float a = fmaxf(o[e], o[e] * slope_out);
                                hi[e] = (__bf16)a;
                                lo[e] = (__bf16)(a - (float)hi[e]);
                            }
                            char* orow = p.ys + row * (size_t)TN * 4 + c8 * 2;
                            *reinterpret_cast<bf16x8*>(orow) = hi;
                            *reinterpret_cast<bf16x8*>(orow + TN * 2) = lo;
                        }
                    }
                }
            }
        }
    };
    if (loader) {
        const int lw = wave - 4;
        const int ltid = tid - 256;
        auto dma_in = [&](const Tile& T) {
            const ConvParams& p = mp.p1[T.b];
            const int pad2 = (mp.p2[T.b].ntaps - 1) >> 1;
            const int R = TMc + p.halo;
            const int ninstr = (R * SPR + 63) >> 6;
            const size_t seq_base = (size_t)T.seq * p.L;
            const int Ls = seq_rows(p, T.seq);
            const int row_bytes = p.cin * 4;
            const int tfirst = T.t0 - pad2 + p.off_min;  // time of LDS row 0
            for (int i = lw; i < ninstr; i += 4) {
                const int n = i * 64 + lane;
                const int r = n >> LOG_SPR;
                const int sl = (n & (SPR - 1)) ^ ((r >> LOG_RPB) & (SPR - 1));
                const int t = tfirst + r;
                const char* src = p.zeros;
                if (r < R && t >= 0 && t < Ls) {
                    if constexpr (F32) src = (p.xf ? reinterpret_cast<const char*>(p.xf) : p.xs) + (seq_base + t) * row_bytes + sl * 16;
                    else src = p.xs + (seq_base + t) * row_bytes + (sl < SPR / 2 ? sl * 16 : p.cin * 2 + (sl - SPR / 2) * 16);
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(smem_b + i * 1024), 16, 0, 2);
            }
        };
        // fp32 input mode: rows are read as fp32, activated and split in registers and written to the same swizzled slots the
        // DMA would fill (logical slot s of row r lives at position s ^ swizzle(r)); out-of-sequence rows become zeros
        auto stage_f32 = [&](const Tile& T) {
            const ConvParams& p = mp.p1[T.b];
            const int pad2 = (mp.p2[T.b].ntaps - 1) >> 1;
            const int R = TMc + p.halo;
            const size_t seq_base = (size_t)T.seq * p.L;
            const int Ls = seq_rows(p, T.seq);
            const int tfirst = T.t0 - pad2 + p.off_min;
            const float slope = p.slope_in;
            constexpr int UPR = CH / 8;  // 8-channel units per row
            constexpr int LOG_UPR = NC16 == 4 ? 3 : 2;
            const int nunits = R * UPR;
            constexpr int UB = 4;        // units in flight per thread: all loads are issued before the first is used
            for (int u0 = ltid; u0 < nunits; u0 += 256 * UB) {
                f32x4 a[UB], b[UB];
#pragma unroll
                for (int q = 0; q < UB; ++q) {
                    const int u = u0 + q * 256;
                    const int r = u >> LOG_UPR;
                    const int t = tfirst + r;
                    a[q] = b[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (u < nunits && t >= 0 && t < Ls) {
                        const float* src = p.xf + (seq_base + t) * CH + (u & (UPR - 1)) * 8;
                        a[q] = *reinterpret_cast<const f32x4*>(src);
                        b[q] = *reinterpret_cast<const f32x4*>(src + 4);
                    }
                }
#pragma unroll
                for (int q = 0; q < UB; ++q) {
                    const int u = u0 + q * 256;
                    if (u < nunits) {
                        const int r = u >> LOG_UPR;
                        const int cu = u & (UPR - 1);
                        const int swz = (r >> LOG_RPB) & (SPR - 1);
                        bf16x8 hi, lo;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float v = e < 4 ? a[q][e] : b[q][e - 4];
                            const float act = fmaxf(v, v * slope);
                            hi[e] = (__bf16)act;
                            lo[e] = (__bf16)(act - (float)hi[e]);
                        }
                        *reinterpret_cast<bf16x8*>(smem_b + r * RB + ((cu ^ swz) << 4)) = hi;
                        *reinterpret_cast<bf16x8*>(smem_b + r * RB + (((SPR / 2 + cu) ^ swz) << 4)) = lo;
                    }
                }
            }
        };
        // exact-fp32 arithmetic: raw fp32 rows (xf) are staged by pure DMA too; the MFMA waves apply LeakyReLU(slope_in) to
        // their conv1 fragments (a handful of VALU ops beside eight 64-cycle MFMAs: free)
        auto stage_in = [&](const Tile& T) {
            if constexpr (F32) {
                dma_in(T);
            } else {
                if (mp.p1[T.b].xf) stage_f32(T);
                else dma_in(T);
            }
        };
        Tile Tprev;
        if (first < my_rounds) stage_in(decode(tile_of(first)));
        for (int it = first, itn; it < my_rounds; it = itn) {
            itn = nxt(it + 1);
            const Tile T = decode(tile_of(it));
            HIFICAR_STAMP(4 * it);
            __syncthreads();                 // A: input of tile `it` landed (hipcc drains vmcnt first)
            HIFICAR_STAMP(4 * it + 1);
            if (it != first) write_out(Tprev, ltid, 256);  // hidden behind conv1 of this tile
            HIFICAR_STAMP(4 * it + 2);
            // F, B, C order LDS only (the out-buffer has been READ: lgkmcnt); the output pass's global stores need not have landed — __syncthreads()
            // would hold the MFMA waves at F for their whole write latency.  The DMA issued behind B is waited for at A (vmcnt(0) there).
            lds_barrier();  // F: shared region free
            lds_barrier();  // B: TS complete, input buffer free
            HIFICAR_STAMP(4 * it + 3);
            if (itn < my_rounds) stage_in(decode(tile_of(itn)));  // hidden behind conv2
            lds_barrier();  // C: conv2 done reading TS
            Tprev = T;
        }
        __syncthreads();                     // Z: the last tile's accumulators are in the out-buffer
        if (first < my_rounds) write_out(Tprev, tid, 512);  // all eight waves share the final output pass
        return;
    }

    // ---------------- MFMA role ----------------
    // The MFMA waves' barriers order LDS only (the loaders wait for their DMA in front of theirs): `s_waitcnt lgkmcnt(0) ; s_barrier` instead of
    // __syncthreads(), whose vmcnt(0) also waits for the weight-ring loads just requested for the next tap — four times per tile.
    auto pair_barrier = [] { lds_barrier(); };
    __builtin_amdgcn_s_setprio(1);  // the MFMA wave outranks the loader wave sharing its SIMD for issue slots
    f32x16 acc[MI];
    using frag_t = typename std::conditional<F32, f32x4, bf16x8>::type;  // 16 bytes per lane either way
    constexpr int NMF = (F32 ? 8 : 3) * MI;  // MFMAs of one K-slab step
    frag_t wr[NC16][2];  // weight ring, one register set per slab, reloaded one tap ahead (see conv_ws_body)
    const frag_t* wp = nullptr;
    int groups_left = 0;
    auto stream1 = [&](const Tile& T) { return reinterpret_cast<const frag_t*>(mp.p1[T.b].w16) + (size_t)wn * mp.p1[T.b].ntaps * NC16 * 128 + lane; };
    auto stream2 = [&](const Tile& T) { return reinterpret_cast<const frag_t*>(mp.p2[T.b].w16) + (size_t)wn * mp.p2[T.b].ntaps * NC16 * 128 + lane; };
    if (first < my_rounds) {
        const Tile T0 = decode(tile_of(first));
        wp = stream1(T0);
#pragma unroll
        for (int u = 0; u < NC16; ++u) {
            wr[u][0] = wp[u * 128];
            wr[u][1] = wp[u * 128 + 64];
        }
        wp += NC16 * 128;
        groups_left = mp.p1[T0.b].ntaps - 1;
    }
    const int wave_row0 = wm * (MI * 32);

    auto addr_set = [&](int buf_off, int r0, int (&ad)[NC16][2]) {
        const int swz = (r0 >> LOG_RPB) & (SPR - 1);
        const int base = buf_off + r0 * RB;
#pragma unroll
        for (int u = 0; u < NC16; ++u) {
            if constexpr (F32) {  // slab u = slots 4u .. 4u+3 (see conv_ws_body)
                ad[u][0] = base + (((4 * u + g) ^ swz) << 4);
                ad[u][1] = base + (((4 * u + 2 + g) ^ swz) << 4);
            } else {
                ad[u][0] = base + (((2 * u + g) ^ swz) << 4);
                ad[u][1] = base + (((SPR / 2 + 2 * u + g) ^ swz) << 4);
            }
        }
    };
    float act_slope = 1.f;  // F32 only: LeakyReLU slope applied to the activation fragments as they are read (1 = none)
    bool act_on = false;
    auto load_x = [&](frag_t (&xh)[MI], frag_t (&xl)[MI], const int (&ad)[2]) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            xh[mi] = *reinterpret_cast<const frag_t*>(smem_b + ad[0] + mi * 32 * RB);
            xl[mi] = *reinterpret_cast<const frag_t*>(smem_b + ad[1] + mi * 32 * RB);
        }
    };
    auto mfma_step = [&](const frag_t (&xh_)[MI], const frag_t (&xl_)[MI], const frag_t& wh, const frag_t& wl) {
        if constexpr (F32) {
            frag_t xh[MI], xl[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                xh[mi] = xh_[mi];
                xl[mi] = xl_[mi];
                if (act_on) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xh[mi][e] = fmaxf(xh[mi][e], xh[mi][e] * act_slope);
                        xl[mi][e] = fmaxf(xl[mi][e], xl[mi][e] * act_slope);
                    }
                }
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(wh[s4], xh[mi][s4], acc[mi], 0, 0, 0);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[s4], xl[mi][s4], acc[mi], 0, 0, 0);
        } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl_[mi], acc[mi], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh_[mi], acc[mi], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh_[mi], acc[mi], 0, 0, 0);
        }
    };
    // one convolution: taps x NC16 K slabs out of the LDS image at buf_off; lane's row for tap t is row0 + t*tap_rows
    auto run_conv = [&](int buf_off, int row0, int tap_rows, int ntaps, const frag_t* wp_next, int groups_next) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
        int ad[NC16][2];
        addr_set(buf_off, row0, ad);
        frag_t x0h[MI], x0l[MI], x1h[MI], x1l[MI];
        load_x(x0h, x0l, ad[0]);
        for (int t = 0; t < ntaps; ++t) {
            const bool last_tap = t + 1 == ntaps;
            int adn[NC16][2];
            addr_set(buf_off, row0 + (last_tap ? t : t + 1) * tap_rows, adn);
            if (groups_left == 0) {  // stream exhausted: continue with the next conv's head
                wp = wp_next;
                groups_left = groups_next;
            }
            --groups_left;
#pragma unroll
            for (int u = 0; u < NC16; u += 2) {
                load_x(x1h, x1l, ad[u + 1]);
                {
                    const frag_t wh = wr[u][0], wl = wr[u][1];
                    wr[u][0] = wp[u * 128];
                    wr[u][1] = wp[u * 128 + 64];
                    mfma_step(x0h, x0l, wh, wl);
                    pin_slab_step<2 * MI, 2, NMF>();
                }
                if (u + 2 < NC16) load_x(x0h, x0l, ad[u + 2]);
                else load_x(x0h, x0l, adn[0]);
                {
                    const frag_t wh = wr[u + 1][0], wl = wr[u + 1][1];
                    wr[u + 1][0] = wp[(u + 1) * 128];
                    wr[u + 1][1] = wp[(u + 1) * 128 + 64];
                    mfma_step(x1h, x1l, wh, wl);
                    pin_slab_step<2 * MI, 2, NMF>();
                }
            }
            wp += NC16 * 128;
#pragma unroll
            for (int u = 0; u < NC16; ++u) {
                ad[u][0] = adn[u][0];
                ad[u][1] = adn[u][1];
            }
        }
    };

    int last = -1;
    for (int it = first, itn; it < my_rounds; it = itn) {
        itn = nxt(it + 1);
        last = it;
        const Tile T = decode(tile_of(it));
        const ConvParams& p1 = mp.p1[T.b];
        const ConvParams& p2 = mp.p2[T.b];
        const int k2 = p2.ntaps;
        const int pad2 = (k2 - 1) >> 1;
        const Tile Tn = decode(tile_of(itn < my_rounds ? itn : it));
        const int Ls = seq_rows(p1, T.seq);
        HIFICAR_STAMP(6 * it);
        pair_barrier();  // A: input landed
        HIFICAR_STAMP(6 * it + 1);
        // ---- conv1 over TMc rows (time t0 - pad2 + r1) ----
        if constexpr (F32) {
            act_on = p1.xf != nullptr;  // raw rows staged: LeakyReLU(slope_in) on the fragments; activated rows (xs): none
            act_slope = p1.slope_in;
        }
        run_conv(0, wave_row0 + li + (p1.tap_off0[0] - p1.off_min), p1.tap_step, p1.ntaps, stream2(T), k2);
        if constexpr (F32) act_on = false;
        HIFICAR_STAMP(6 * it + 2);
        pair_barrier();  // F: the loaders are done with the previous tile's out-buffer (same LDS region as TS)
        {   // bias + LeakyReLU + split -> TS (zero outside the sequence: conv2's padding)
            const float slope = mp.slope_mid;
            f32x4 bias4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bias4[q] = *reinterpret_cast<const f32x4*>(p1.bias + wn * 32 + 8 * q + 4 * g);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int r1 = wave_row0 + mi * 32 + li;
                const int t = T.t0 - pad2 + r1;
                const bool in_seq = t >= 0 && t < Ls;
                const int swz = (r1 >> LOG_RPB) & (SPR - 1);
                char* trow = smem_b + ts_off + r1 * RB + 8 * g;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if constexpr (F32) {
                        f32x4 a4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = acc[mi][4 * q + e] + bias4[q][e];
                            a4[e] = in_seq ? fmaxf(v, v * slope) : 0.f;
                        }
                        const int slot = wn * 8 + 2 * q + g;  // this lane's 4 channels 32*wn + 8q + 4g .. +3 are one 16-byte slot
                        *reinterpret_cast<f32x4*>(smem_b + ts_off + r1 * RB + ((slot ^ swz) << 4)) = a4;
                    } else {
                        bf16x4 hi, lo;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = acc[mi][4 * q + e] + bias4[q][e];
                            const float a = in_seq ? fmaxf(v, v * slope) : 0.f;
                            hi[e] = (__bf16)a;
                            lo[e] = (__bf16)(a - (float)hi[e]);
                        }
                        const int slot = wn * 4 + q;  // 8-channel group of this lane's 4 channels
                        *reinterpret_cast<bf16x4*>(trow + ((slot ^ swz) << 4)) = hi;
                        *reinterpret_cast<bf16x4*>(trow + (((SPR / 2 + slot) ^ swz) << 4)) = lo;
                    }
                }
            }
        }
        HIFICAR_STAMP(6 * it + 3);
        pair_barrier();  // B: input buffer free, TS complete
        HIFICAR_STAMP(6 * it + 4);
        // ---- conv2 over TS: output row r2 reads TS rows r2 .. r2 + k2 - 1 ----
        run_conv(ts_off, wave_row0 + li, 1, k2, stream1(Tn), mp.p1[Tn.b].ntaps);
        HIFICAR_STAMP(6 * it + 5);
        pair_barrier();  // C: every wave is done reading TS; its region becomes the out-buffer
        {
            float* O = reinterpret_cast<float*>(smem_b + o_off);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[mi][4 * q + e];
                    *reinterpret_cast<f32x4*>(&O[(wave_row0 + mi * 32 + li) * OP + wn * 32 + 8 * q + 4 * g]) = v;
                }
        }
    }
    pair_barrier();  // Z
    if (last >= 0) write_out(decode(tile_of(last)), tid, 512);
}

template <int MI, int WM, int WN, int NC16>
__global__ __launch_bounds__(512) void conv_pair_bf16x3_kernel(const PairParams mp) {
    conv_pair_body<MI, WM, WN, NC16, false>(mp);
}

// Exact-fp32 arithmetic of the fused pair: fp32 rows staged by pure DMA (raw residual-stream rows: LeakyReLU is applied to the
// conv1 fragments by the MFMA waves; no activated copy exists anywhere), fp32 intermediate in LDS, v_mfma_f32_32x32x2_f32.
template <int MI, int WM, int WN, int NC16>
__global__ __launch_bounds__(512) void conv_pair_f32_kernel(const PairParams mp) {
    conv_pair_body<MI, WM, WN, NC16, true>(mp);
}

}  // namespace hificar
