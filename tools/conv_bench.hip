// Developer tool: launches the real bf16x3 wave-specialised conv kernel on a stage-shaped problem, times it and
// dumps one workgroup's timeline (s_memtime stamps).  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHIFICAR_TRACE tools/conv_bench.hip -o tools/conv_bench.bin
#include "r05_kernels/hificar_kernels_r05.hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
using namespace hificar;

static void summarize(const std::vector<unsigned long long>& ht, int G) {
    // per-workgroup timeline summary over ALL workgroups (s_memtime ticks; MFMA wave 0 = row 0, loader wave 4 = row 1)
    auto stat = [&](const char* what, std::vector<double> v) {
        if (v.empty()) return;
        std::sort(v.begin(), v.end());
        printf("    %-34s min %8.0f  med %8.0f  p90 %8.0f  max %8.0f\n", what, v.front(), v[v.size() / 2], v[v.size() * 9 / 10], v.back());
    };
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < G; ++w) t0 = std::min(t0, ht[(size_t)w * 128]);
    std::vector<double> start, first_ready, mfma_end, wg_end, busy;
    for (int w = 0; w < G; ++w) {
        const unsigned long long* m = &ht[(size_t)w * 128];
        start.push_back((double)(m[0] - t0));
        if (m[2]) first_ready.push_back((double)(m[2] - m[0]));  // first barrier released: first item staged
        if (m[62]) mfma_end.push_back((double)(m[62] - t0));
        if (m[63]) wg_end.push_back((double)(m[63] - t0));
        if (m[62] && m[2]) busy.push_back((double)(m[62] - m[2]));
    }
    stat("WG start (vs first WG)", start);
    stat("start -> first item staged", first_ready);
    stat("MFMA phase (first item -> last acc)", busy);
    stat("last accumulators ready (abs)", mfma_end);
    stat("WG end (abs)", wg_end);
}

template <int MI, int WM, int WN, int NC16, bool F32 = false, int KS = 1, bool DO = false>
void run(const char* label, int nseq, int L, int C, int nbr, const int* ks, int dil, bool residual, int mode = 0) {
    constexpr int TM = WM * MI * 32;
    const int CH = NC16 * 16;
    float *x, *y[3], *bias;
    char *xs, *ys[3], *zeros;
    const size_t n = (size_t)nseq * L * C;
    hipMalloc(&x, n * 4);
    hipMalloc(&xs, n * 4);
    static const bool zero_data = getenv("CONV_BENCH_ZERO") != nullptr;    // all-zero operands: no toggling -> the clock DVFS allows
    static const bool rand_data = getenv("CONV_BENCH_RAND") != nullptr;    // random activations and weights (realistic toggling)
    hipMemset(xs, zero_data ? 0 : 0x3c, n * 4);
    if (rand_data) {
        std::vector<float> hr(n);
        unsigned st = 12345u;
        for (size_t i = 0; i < n; ++i) { st = st * 1664525u + 1013904223u; hr[i] = ((st >> 8) & 0xffff) / 32768.f - 1.f; }
        if (!F32) {  // split rows hold bf16 pairs: random bf16 bit patterns of moderate magnitude
            uint16_t* hb = reinterpret_cast<uint16_t*>(hr.data());
            for (size_t i = 0; i < 2 * n; ++i) { st = st * 1664525u + 1013904223u; hb[i] = 0x3c00 + ((st >> 10) & 0x1ff) + ((st >> 3) & 0x8000); }
        }
        hipMemcpy(xs, hr.data(), n * 4, hipMemcpyHostToDevice);
    }
    hipMalloc(&zeros, 256);
    hipMemset(zeros, 0, 256);
    hipMalloc(&bias, C * 4);
    hipMemset(bias, 0, C * 4);
    std::vector<float> hx(n);
    for (size_t i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    MultiConvParams mp;
    memset(&mp, 0, sizeof(mp));
    mp.zrep = 1;
    int max_halo = 0;
    double flops = 0;
    for (int b = 0; b < nbr; ++b) {
        hipMalloc(&y[b], n * 4);
        hipMemset(y[b], 0, n * 4);
        hipMalloc(&ys[b], n * 4);
        const int K = ks[b], pad = (K - 1) / 2 * dil;
        const size_t wel = ((size_t)(C / 32) * (C / 16) * K * 2 + 2 * NC16) * 512;
        uint16_t* w16;
        hipMalloc(&w16, wel * 2);
        std::vector<uint16_t> hw(wel);
        for (size_t i = 0; i < wel; ++i) hw[i] = 0x3c00 + (uint16_t)((i * 40503u) & 0xff);
        if (F32) {  // fp32 fragments: small random values in the same buffer
            float* hf = reinterpret_cast<float*>(hw.data());
            unsigned s = 777u + b;
            for (size_t i = 0; i < wel / 2; ++i) { s = s * 1664525u + 1013904223u; hf[i] = ((int)(s >> 9) - (1 << 22)) * (0.05f / (1 << 22)); }
        }  // small bf16 values
        if (zero_data) std::fill(hw.begin(), hw.end(), 0);
        if (rand_data) {
            unsigned st = 777u + b;
            if (F32) { float* hf = reinterpret_cast<float*>(hw.data()); for (size_t i = 0; i < wel / 2; ++i) { st = st * 1664525u + 1013904223u; hf[i] = (((st >> 8) & 0xffff) / 32768.f - 1.f) * 0.05f; } }
            else for (size_t i = 0; i < wel; ++i) { st = st * 1664525u + 1013904223u; hw[i] = 0x3a00 + ((st >> 10) & 0x1ff) + ((st >> 3) & 0x8000); }
        }
        hipMemcpy(w16, hw.data(), wel * 2, hipMemcpyHostToDevice);
        ConvParams& p = mp.p[b];
        p.len_const = -1;
        p.w16 = reinterpret_cast<const bf16x8*>(w16); p.bias = bias; p.res = residual ? x : nullptr; p.y = residual ? y[b] : nullptr;
        p.xs = xs; p.ys = (mode & 1) ? nullptr : ys[b]; if (mode & 2) p.y = y[b]; p.zeros = zeros; p.slope_out = 0.1f; p.cout_real = C;
        p.L = L; p.tiles_per_seq = (L + TM - 1) / TM; p.cin = C; p.cout_total = C;
        p.n_blocks32 = C / 32; p.nb32_per_phase = C / 32; p.ntaps = K; p.off_min = -pad; p.halo = 2 * pad;
        p.tap_step = dil; p.tap_off0[0] = -pad;
        max_halo = std::max(max_halo, p.halo);
        flops += 2.0 * nseq * L * (double)C * C * K;
    }
    mp.n_branches = nbr;
    mp.nseq_tiles = nseq * ((L + TM - 1) / TM);
    mp.ngroups = (C / 32 + WN - 1) / WN;
    mp.total_tiles = nbr * mp.ngroups * mp.nseq_tiles;
    mp.buf_bytes = ((TM + max_halo) * (CH * 4) + 1023) / 1024 * 1024;
    const int G = std::min(mp.total_tiles, 256);
    unsigned long long* trace;
    hipMalloc(&trace, (size_t)G * 2 * 64 * 8);
    hipMemset(trace, 0, (size_t)G * 2 * 64 * 8);
    mp.trace = trace;
    void (*kern)(const MultiConvParams) = nullptr;
    if constexpr (DO) kern = conv_f32do_kernel<MI, WM, WN, NC16>;  // direct output (round 4): no out-buffer
    else if constexpr (KS == 4) kern = F32 ? conv_sk_f32_kernel<MI, NC16> : conv_sk_bf16x3_kernel<MI, NC16>;  // split-K form (WM = WN = 1)
    else kern = F32 ? conv_f32_kernel<MI, WM, WN, NC16> : conv_bf16x3_kernel<MI, WM, WN, NC16>;
    const size_t lds_bytes = 2 * (size_t)mp.buf_bytes + (DO ? 0 : (size_t)KS * TM * (WN * 32 + 4) * 4);
    const int nthreads = KS == 4 ? 512 : (WM * WN + 4) * 64;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0, ms1 = 0;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, dim3(G), dim3(nthreads), lds_bytes, 0, mp);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemset(trace, 0, (size_t)G * 2 * 64 * 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(G), dim3(nthreads), lds_bytes, 0, mp);  // the traced launch: alone
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms1, e0, e1);
    printf("%-28s %s tiles=%d G=%d lds=%dKB  %.1f us/launch (back to back), %.1f us alone  %.0f TF-alg\n", label, F32 ? "f32" : "bf16x3", mp.total_tiles, G,
           2 * mp.buf_bytes / 1024, ms * 100, ms1 * 1000, flops / (ms * 1e-4) / 1e12);
    std::vector<unsigned long long> ht((size_t)G * 2 * 64);
    hipMemcpy(ht.data(), trace, ht.size() * 8, hipMemcpyDeviceToHost);
    summarize(ht, G);
    for (int wg : {0, G / 2}) {
        const unsigned long long* m = &ht[(size_t)wg * 2 * 64];
        const unsigned long long* l = m + 64;
        const unsigned long long t0 = std::min(m[0], l[0]);
        printf("  WG %d MFMA : ", wg);
        for (int i = 0; i < 40; ++i) printf("%lld ", m[i] ? (long long)(m[i] - t0) / 100 : -1LL);
        printf("\n  WG %d load : ", wg);
        for (int i = 0; i < 30; ++i) printf("%lld ", l[i] ? (long long)(l[i] - t0) / 100 : -1LL);
        printf("  (x100 clk)\n");
    }
}

template <int MI, int WM, int WN, int NC16, bool F32 = false>
void run_pair(const char* label, int nseq, int L, int C, int nbr, const int* ks, int dil) {
    constexpr int TMc = WM * MI * 32;
    float *x, *y[3], *bias;
    char *xs, *ys[3], *zeros;
    const size_t n = (size_t)nseq * L * C;
    hipMalloc(&x, n * 4); hipMemset(x, 0, n * 4);
    hipMalloc(&xs, n * 4); hipMemset(xs, 0x3c, n * 4);
    if (F32) {  // exact-fp32 form: random fp32 rows (the clock follows the operands' entropy)
        std::vector<float> hx(n);
        unsigned s = 12345u;
        for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; hx[i] = ((int)(s >> 9) - (1 << 22)) * (1.0f / (1 << 22)); }
        if (getenv("CONV_BENCH_PAIR_ZERO")) std::fill(hx.begin(), hx.end(), 0.f);  // all-zero activations: same instruction stream, same cycle count
        hipMemcpy(xs, hx.data(), n * 4, hipMemcpyHostToDevice);
        hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    }
    hipMalloc(&zeros, 256); hipMemset(zeros, 0, 256);
    hipMalloc(&bias, C * 4); hipMemset(bias, 0, C * 4);
    PairParams pp;
    memset(&pp, 0, sizeof(pp));
    int halo_max = 0, tile = 0;
    double flops = 0;
    for (int b = 0; b < nbr; ++b) {
        hipMalloc(&y[b], n * 4); hipMalloc(&ys[b], n * 4);
        const int K = ks[b], pad = (K - 1) / 2 * dil;
        const size_t wel = ((size_t)(C / 32) * (C / 16) * K * 2 + 4 * NC16) * 512;
        uint16_t *w1, *w2;
        hipMalloc(&w1, wel * 2); hipMalloc(&w2, wel * 2);
        std::vector<uint16_t> hw(wel);
        for (size_t i = 0; i < wel; ++i) hw[i] = 0x3c00 + (uint16_t)((i * 40503u) & 0xff);
        if (F32) {  // fp32 fragments: small random values in the same buffer
            float* hf = reinterpret_cast<float*>(hw.data());
            unsigned s = 777u + b;
            for (size_t i = 0; i < wel / 2; ++i) { s = s * 1664525u + 1013904223u; hf[i] = ((int)(s >> 9) - (1 << 22)) * (0.05f / (1 << 22)); }
        }
        hipMemcpy(w1, hw.data(), wel * 2, hipMemcpyHostToDevice);
        hipMemcpy(w2, hw.data(), wel * 2, hipMemcpyHostToDevice);
        ConvParams& p = pp.p1[b];
        p.len_const = -1;
        pp.p2[b].len_const = -1;
        p.xs = xs; p.w16 = reinterpret_cast<const bf16x8*>(w1); p.bias = bias; p.zeros = zeros;
        p.L = L; p.cin = C; p.cout_total = C; p.n_blocks32 = C / 32; p.nb32_per_phase = C / 32; p.ntaps = K; p.off_min = -pad; p.halo = 2 * pad;
        p.tap_step = dil; p.tap_off0[0] = -pad;
        ConvParams& q = pp.p2[b];
        q.w16 = reinterpret_cast<const bf16x8*>(w2); q.bias = bias; q.res = x; q.y = y[b]; q.ys = ys[b]; q.slope_out = 0.1f;
        q.L = L; q.cin = C; q.cout_total = C; q.cout_real = C; q.ntaps = K; q.n_blocks32 = C / 32;
        halo_max = std::max(halo_max, p.halo);
        const int tmo = TMc - (K - 1);
        pp.tiles_per_seq[b] = (L + tmo - 1) / tmo;
        pp.tile_start[b] = tile;
        tile += nseq * pp.tiles_per_seq[b];
        flops += 2.0 * nseq * L * (double)C * C * K * 2;
    }
    pp.tile_start[nbr] = tile; pp.n_branches = nbr; pp.nseq = nseq;
    pp.in_bytes = ((TMc + halo_max) * C * 4 + 1023) / 1024 * 1024;
    pp.ts_bytes = std::max((TMc + 16) * C * 4, TMc * (C + 4) * 4);
    pp.slope_mid = 0.1f;
    const int G = std::min(tile, 256);
    unsigned long long* trace;
    hipMalloc(&trace, (size_t)G * 2 * 64 * 8); hipMemset(trace, 0, (size_t)G * 2 * 64 * 8);
    pp.trace = trace;
    void (*kern)(const PairParams) = F32 ? conv_pair_f32_kernel<MI, WM, WN, NC16> : conv_pair_bf16x3_kernel<MI, WM, WN, NC16>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t lds = pp.in_bytes + pp.ts_bytes;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, dim3(G), dim3(512), lds, 0, pp);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-28s %s tiles=%d G=%d lds=%zuKB  %.1f us/launch(pair)  %.0f TF-alg\n", label, F32 ? "f32" : "bf16x3", tile, G, lds / 1024, ms * 100, flops / (ms * 1e-4) / 1e12);
    std::vector<unsigned long long> ht((size_t)G * 2 * 64);
    hipMemcpy(ht.data(), trace, ht.size() * 8, hipMemcpyDeviceToHost);
    {   // where the two roles spend a tile, summed over all workgroups and tiles (s_memtime ticks): MFMA waves: wait A | conv1 | wait F + epilogue |
        // wait B | conv2 | wait C + hand-over;  loader waves: wait A | output pass | wait F + B | staging + wait C
        double mf[6] = {0, 0, 0, 0, 0, 0}, ld[4] = {0, 0, 0, 0};
        for (int w = 0; w < G; ++w) {
            const unsigned long long* m = &ht[(size_t)w * 128];
            const unsigned long long* l = m + 64;
            for (int it = 0; 6 * it + 5 < 60 && m[6 * it + 5]; ++it) {
                for (int q = 0; q < 5; ++q) mf[q] += (double)(m[6 * it + q + 1] - m[6 * it + q]);
                if (6 * it + 6 < 60 && m[6 * it + 6]) mf[5] += (double)(m[6 * it + 6] - m[6 * it + 5]);
            }
            for (int it = 0; 4 * it + 3 < 44 && l[4 * it + 3]; ++it) {
                for (int q = 0; q < 3; ++q) ld[q] += (double)(l[4 * it + q + 1] - l[4 * it + q]);
                if (4 * it + 4 < 44 && l[4 * it + 4]) ld[3] += (double)(l[4 * it + 4] - l[4 * it + 3]);
            }
        }
        const double mt = mf[0] + mf[1] + mf[2] + mf[3] + mf[4] + mf[5], lt = ld[0] + ld[1] + ld[2] + ld[3];
        printf("  MFMA waves: wait A %.1f%% | conv1 %.1f%% | wait F + epilogue %.1f%% | wait B %.1f%% | conv2 %.1f%% | wait C + hand-over %.1f%%\n",
               100 * mf[0] / mt, 100 * mf[1] / mt, 100 * mf[2] / mt, 100 * mf[3] / mt, 100 * mf[4] / mt, 100 * mf[5] / mt);
        printf("  loaders   : wait A %.1f%% | output pass %.1f%% | wait F + B %.1f%% | staging + wait C %.1f%%\n", 100 * ld[0] / lt, 100 * ld[1] / lt,
               100 * ld[2] / lt, 100 * ld[3] / lt);
    }
    for (int wg : {0}) {
        const unsigned long long* m = &ht[(size_t)wg * 2 * 64];
        const unsigned long long* l = m + 64;
        const unsigned long long t0 = std::min(m[0] ? m[0] : ~0ull, l[0] ? l[0] : ~0ull);
        printf("  WG %d MFMA [A? A conv1 epi B conv2]x : ", wg);
        for (int i = 0; i < 60; ++i) printf("%lld ", m[i] ? (long long)(m[i] - t0) / 100 : -1LL);
        printf("\n  WG %d load [A? A wout B]x : ", wg);
        for (int i = 0; i < 44; ++i) printf("%lld ", l[i] ? (long long)(l[i] - t0) / 100 : -1LL);
        printf("\n");
    }
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "pairf32")) {  // the fused conv1 -> conv2 kernel in exact fp32: where the two roles spend a tile
        const int k3[3] = {11, 7, 3};
        run_pair<4, 2, 2, 4, true>("PAIR stage2 C64 L1000", 64, 1000, 64, 3, k3, 1);
        run_pair<4, 2, 2, 4, true>("PAIR stage2 C64 L984 (4 full tiles)", 64, 984, 64, 3, k3, 1);
        run_pair<4, 4, 1, 2, true>("PAIR stage3 C32 L2000", 64, 2000, 32, 3, k3, 1);
        run_pair<4, 2, 2, 4>("PAIR stage2 C64 L1000", 64, 1000, 64, 3, k3, 1);
        run_pair<4, 4, 1, 2>("PAIR stage3 C32 L2000", 64, 2000, 32, 3, k3, 1);
        return 0;
    }
    {
        const int k3[3] = {11, 7, 3};
        run_pair<4, 2, 2, 4>("PAIR stage2 C64 L1000", 64, 1000, 64, 3, k3, 1);
        run_pair<4, 4, 1, 2>("PAIR stage3 C32 L2000", 64, 2000, 32, 3, k3, 1);
    }
    const int k3[3] = {11, 7, 3};
    if (argc > 1 && !strcmp(argv[1], "small")) {  // the small-batch launches of the AR loop: split-K form, exact fp32
        for (int B : {1, 8}) {
            printf("---- batch %d\n", B);
            run<1, 1, 1, 4, true, 4>("stage0 sk<1,4> conv1", B, 125, 256, 3, k3, 1, false);
            run<1, 1, 1, 4, true, 4>("stage0 sk<1,4> conv2+res", B, 125, 256, 3, k3, 1, true);
            run<2, 1, 1, 4, true, 4>("stage0 sk<2,4> conv1", B, 125, 256, 3, k3, 1, false);
            run<1, 1, 1, 4, true, 4>("stage1 sk<1,4> conv1", B, 500, 128, 3, k3, 1, false);
            run<1, 1, 4, 4, true>("stage1 <1,1,4,4> conv1", B, 500, 128, 3, k3, 1, false);
            run<1, 1, 1, 2, true, 4>("stage2 sk<1,2> conv1", B, 1000, 64, 3, k3, 1, false);
            run<1, 2, 2, 2, true>("stage2 <1,2,2,2> conv1", B, 1000, 64, 3, k3, 1, false);
            run<1, 1, 1, 1, true, 4>("stage3 sk<1,1> conv1", B, 2000, 32, 3, k3, 1, false);
            run<1, 4, 1, 1, true>("stage3 <1,4,1,1> conv1", B, 2000, 32, 3, k3, 1, false);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "f32do")) {  // direct-output kernels: the narrow stages, and C = 64 with ONE 64-channel chunk per tile
        run<4, 1, 4, 4, true, 1, true>("stage0 (4,1,4) conv2+res DO", 64, 125, 256, 3, k3, 1, true);
        run<4, 2, 2, 2, true, 1, true>("stage2 C64 (4,2,2) conv1 DO", 64, 1000, 64, 3, k3, 1, false);
        run<4, 2, 2, 2, true, 1, true>("stage2 C64 (4,2,2) conv2+res DO", 64, 1000, 64, 3, k3, 1, true);
        run<4, 2, 2, 4, true, 1, true>("stage2 C64 chunk64 conv1 DO", 64, 1000, 64, 3, k3, 1, false);
        run<4, 2, 2, 4, true, 1, true>("stage2 C64 chunk64 conv2+res DO", 64, 1000, 64, 3, k3, 1, true);
        run<4, 4, 1, 1, true, 1, true>("stage3 C32 (4,4,1) conv1 DO", 64, 2000, 32, 3, k3, 1, false);
        run<4, 4, 1, 1, true, 1, true>("stage3 C32 (4,4,1) conv2+res DO", 64, 2000, 32, 3, k3, 1, true);
        run<4, 4, 1, 2, true, 1, true>("stage3 C32 chunk32 conv1 DO", 64, 2000, 32, 3, k3, 1, false);
        run<4, 4, 1, 2, true, 1, true>("stage3 C32 chunk32 conv2+res DO", 64, 2000, 32, 3, k3, 1, true);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "f32")) {
        run<2, 2, 4, 4, true>("stage0 8w (2,2,4) conv1", 64, 125, 256, 3, k3, 1, false);
        run<2, 2, 4, 4, true>("stage0 8w (2,2,4) conv2+res", 64, 125, 256, 3, k3, 1, true);
        run<1, 2, 4, 4, true>("stage0 8w (1,2,4) conv1", 64, 125, 256, 3, k3, 1, false);
        run<2, 2, 4, 4, true>("stage1 8w (2,2,4) conv1", 64, 500, 128, 3, k3, 1, false);
        run<2, 2, 4, 4, true>("stage1 8w (2,2,4) conv2+res", 64, 500, 128, 3, k3, 1, true);
        run<2, 4, 2, 2, true>("stage2 8w C64 (2,4,2) conv1", 64, 1000, 64, 3, k3, 1, false);
        run<2, 4, 2, 2, true>("stage2 8w C64 (2,4,2) conv2+res", 64, 1000, 64, 3, k3, 1, true);
        run<2, 8, 1, 1, true>("stage3 8w C32 (2,8,1) conv1", 64, 2000, 32, 3, k3, 1, false);
        run<2, 8, 1, 1, true>("stage3 8w C32 (2,8,1) conv2+res", 64, 2000, 32, 3, k3, 1, true);
        run<2, 1, 4, 4, true>("stage0 (2,1,4) conv1", 64, 125, 256, 3, k3, 1, false);
        run<2, 1, 4, 4, true>("stage0 (2,1,4) conv2+res", 64, 125, 256, 3, k3, 1, true);
        run<2, 1, 4, 4, true>("stage0 (2,1,4) conv1 B=128", 128, 125, 256, 3, k3, 1, false);
        run<4, 1, 4, 4, true>("stage0 (4,1,4) conv1", 64, 125, 256, 3, k3, 1, false);
        run<4, 1, 4, 4, true>("stage1 (4,1,4) conv1", 64, 500, 128, 3, k3, 1, false);
        run<4, 1, 4, 4, true>("stage1 (4,1,4) conv2+res", 64, 500, 128, 3, k3, 1, true);
        run<4, 1, 4, 4, true>("stage1 (4,1,4) conv1 B=128", 128, 500, 128, 3, k3, 1, false);
        run<4, 2, 2, 2, true>("stage2 C64 (4,2,2) conv1", 64, 1000, 64, 3, k3, 1, false);
        run<4, 2, 2, 2, true>("stage2 C64 (4,2,2) conv2+res", 64, 1000, 64, 3, k3, 1, true);
        run<4, 2, 2, 2, true>("stage2 C64 conv1 B=128", 128, 1000, 64, 3, k3, 1, false);
        run<4, 4, 1, 1, true>("stage3 C32 (4,4,1) conv1", 64, 2000, 32, 3, k3, 1, false);
        run<4, 4, 1, 1, true>("stage3 C32 (4,4,1) conv2+res", 64, 2000, 32, 3, k3, 1, true);
        run<4, 4, 1, 1, true>("stage3 C32 conv1 B=128", 128, 2000, 32, 3, k3, 1, false);
        return 0;
    }
    run<2, 2, 4, 4>("stage0 8w (2,2,4) conv1", 64, 125, 256, 3, k3, 1, false);
    run<2, 2, 4, 4>("stage0 8w (2,2,4) conv2+res", 64, 125, 256, 3, k3, 1, true);
    run<2, 2, 4, 4>("stage1 8w (2,2,4) conv1", 64, 500, 128, 3, k3, 1, false);
    run<2, 2, 4, 4>("stage1 8w (2,2,4) conv2+res", 64, 500, 128, 3, k3, 1, true);
    run<4, 1, 4, 4>("stage0 TM128 TN128 (4,1,4)", 64, 125, 256, 3, k3, 1, false);
    run<2, 1, 4, 4>("stage0 TM64 TN128 (2,1,4)", 64, 125, 256, 3, k3, 1, false);
    run<2, 2, 2, 4>("stage0 TM128 TN64 (2,2,2)", 64, 125, 256, 3, k3, 1, false);
    run<2, 2, 2, 4>("stage0 (2,2,2) conv2+res", 64, 125, 256, 3, k3, 1, true);
    run<4, 1, 4, 4>("stage1 (4,1,4) conv1", 64, 500, 128, 3, k3, 1, false);
    run<4, 1, 4, 4>("stage1 (4,1,4) conv2+res", 64, 500, 128, 3, k3, 1, true);
    run<4, 2, 2, 2>("stage2 C64 L1000 (4,2,2) conv1", 64, 1000, 64, 3, k3, 1, false);
    run<4, 2, 2, 2>("stage2 C64 (4,2,2) conv2+res", 64, 1000, 64, 3, k3, 1, true);
    run<4, 4, 1, 1>("stage3 C32 L2000 (4,4,1) conv1", 64, 2000, 32, 3, k3, 1, false);
    run<4, 4, 1, 1>("stage3 C32 (4,4,1) conv2+res", 64, 2000, 32, 3, k3, 1, true);
    return 0;
}
