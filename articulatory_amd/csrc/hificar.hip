// libhificar.so — host side of the C ABI declared in include/hificar.h.
// Builds the layer list of the reference's HiFiGANGenerator (articulatory/models/hifigan.py:108-175)
// from a hificar_config, repacks folded weights into the kernels' layouts, plans the workspace and
// enqueues the forward pass / the batched autoregressive loop on the caller's HIP stream.
#include "hificar_kernels.hip.h"
#include "hificar_launch.h"
#include "hificar_backward.hip.h"
#include "hificar_disc_kernels.hip.h"

#include "../../include/hificar.h"

#include <algorithm>
#include <queue>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace hificar;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return fail(HIFICAR_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// model description
// ------------------------------------------------------------------------------------------------
struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

struct ConvLayer {
    std::string name;     // reference module path, e.g. "blocks.3.convs1.2.1"
    bool transposed = false;
    int cin = 0;          // real input channels
    int cin_pad = 0;      // row pitch of the input buffer (multiple of 16)
    int cout = 0;         // real output channels (per phase for transposed)
    int cout_pad = 0;     // output channels per phase as laid out in memory: cout rounded up to a multiple of 32 (extra channels have
                          // zero weights and bias, so they hold exact zeros through every layer and cost nothing in accuracy)
    int K = 0, dilation = 1, stride = 1, padding = 0;
    bool has_bias = true;
    // derived
    int n_phase = 1, ntaps = 0, cout_total = 0;
    int off_min = 0, off_max = 0;
    int tap_off[kMaxPhase][kMaxTaps];
    int tap_k[kMaxPhase][kMaxTaps];  // which kernel index each (phase, tap) uses; -1 = zero weights
    float* d_bias = nullptr;
    int chunk16 = 0, n_blocks32 = 0, nb32_per_phase = 0;
    uint16_t* d_w16 = nullptr;   // bf16x3 arithmetic: hi/lo bf16 weight fragments in MFMA lane order
    uint16_t* d_w16c = nullptr;  // same fragments packed with one K chunk = all channels (fused pair kernel, C <= 64)
    float* d_w32 = nullptr;      // exact-fp32 arithmetic: fp32 fragments in the same order
    float* d_w32c = nullptr;     // fp32 fragments with one K chunk = all channels (fused pair kernel, C <= 64)
};

// One GBlock of a GBlockGenerator (articulatory/layers/pytorch_layers.py:32-91): conv1 = [ReLU, Upsample, c1a, ReLU, c1b (dilation 3)],
// res1 = [Upsample, res (1 x 1)], conv2 = [ReLU, c2a (dilation 9), ReLU, c2b (dilation 27)]
struct GBlockLayers {
    ConvLayer c1a, c1b, res, c2a, c2b;
    int scale = 1;
    int cin = 0, cout = 0;
};

struct hificar_handle {
    hificar_config cfg;
    int arch = 0;                  // 0: HiFiGANGenerator (hificar_create); 1: GBlockGenerator (hificar_gblock_create, hificar_gblock.hip.inc)
    std::vector<GBlockLayers> gb;  // arch 1
    int c_last = 0;                // channels in front of the output conv
    bool finalized = false;
    int precision = HIFICAR_PREC_F32;
    bool profile_detail = false;   // HIFICAR_PROFILE_DETAIL=1: profile rows carry the layer name
    bool use_pair = true;          // HIFICAR_PAIR=0: run narrow stages layer by layer (A/B runs)
    bool use_lpt = true;           // false: round-robin tile walk instead of the host LPT schedule
    double mi1_penalty = 1.05;     // cost factor of 32-row tiles in the exact-fp32 tile choice (they re-stream the weights most often: L2-bound when
                                   // K is long).  The discriminator engine raises it: its launches overlap on several streams, so a nearly
                                   // empty last round of taller tiles costs little there, while the L2 traffic of short tiles is shared by all
    int pick_throughput = 0;       // > 0: launches of at least this many tiles choose their tile shape by workgroup-time instead of makespan (set by
                                   // the discriminator engine, whose sub-networks run on eight streams; HIFICAR_DISC_PICK overrides, 0 = off)
    bool xcd_order = true;         // XCD-contiguous tile order for one-round launches that stream more weights than activations
    bool pair_small = true;        // HIFICAR_PAIR_SMALL: 128-row fused pair tiles at C = 32 for mid-size launches (pair_small_tiles)
    int ksplit = 1;                // HIFICAR_KSPLIT: 0 = never use the split-K conv form, 1 = when it is estimated faster (default), 2 = always
    int cf = 0;       // feature channels = in_channels - ar_output*use_ar
    int cin_pad = 0;  // padded input-conv channels
    int hop = 1;
    int num_cus = 256;
    std::map<std::string, std::vector<int64_t>> expected;  // name -> shape
    std::map<std::string, HostTensor> tensors;
    ConvLayer input_conv;
    std::vector<ConvLayer> ups;
    std::vector<ConvLayer> convs1;  // [stage][block][dil] flattened
    std::vector<ConvLayer> convs2;
    // output conv
    float* d_out_w = nullptr;
    float out_bias = 0.f;
    float* d_out_bias = nullptr;   // training (device-resident weights): the output conv's bias is read from here
    void* train = nullptr;         // TrainState (hificar_train.hip.inc), created by the first hificar_set_weight_device
    void (*train_free)(hificar_handle*) = nullptr;
    // MLP
    float* d_mlp_w[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    float* d_mlp_b[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    // speaker / phoneme conditioning
    float* d_spk_emb = nullptr;
    float* d_spk_w = nullptr;
    float* d_spk_b = nullptr;
    float* d_ph_emb = nullptr;
    float* d_phfc_w = nullptr;
    float* d_phfc_b = nullptr;
    std::vector<void*> allocs;
    char* d_zeros = nullptr;  // 256 bytes of zeros: source of padding rows for the LDS DMA
    void* d_tab = nullptr;    // step table of hificar_ar_loop_packed (grown on demand) and its pinned host staging copy
    void* h_tab = nullptr;
    size_t tab_bytes = 0;
    hipEvent_t tab_copied = nullptr;  // recorded behind the last upload: the staging copy may be rewritten once it has fired
    // tile schedules (LPT assignment of tiles to workgroups), cached per launch shape.  They live in append-only arenas: a
    // device block plus a pinned host mirror, filled on the host and uploaded with ONE hipMemcpyAsync on the launch stream, so
    // the first use of a new launch shape neither allocates nor synchronises (the first arena is allocated in hificar_finalize)
    struct Sched {
        int* d_start = nullptr;
        int* d_tiles = nullptr;
    };
    std::map<std::string, Sched> scheds;
    std::map<std::string, int> tile_picks;  // launch shape -> index into kTileCfgs (launch_conv's choice, cached: it simulates the LPT assignment)
    struct Arena {
        char* d = nullptr;
        char* h = nullptr;
        size_t cap = 0, used = 0;
    };
    std::vector<Arena> arenas;
    // device copies of the batched-reduction tables of the backward passes (flush_reduce, hificar_train.hip.inc), cached by content: in steady
    // state a training iteration re-uses the tables of the iteration before (same buffers from the caller's caching allocator) without any upload
    struct ReduceSlot {
        char* d = nullptr;
        char* h = nullptr;
        size_t bytes = 0;
        unsigned long long hash = 0, stamp = 0;
        hipEvent_t ev = nullptr;  // recorded behind the last launch that reads the slot
        hipEvent_t up = nullptr;  // recorded behind the upload
        hipStream_t up_stream = nullptr;
    };
    std::vector<ReduceSlot> rslots;
    unsigned long long rstamp = 0;
    // every call's work is ordered behind the previous call's even when the caller switches streams (the schedules, the packed
    // step table and the workspace are shared state of the handle)
    hipStream_t last_stream = nullptr;
    bool have_last_stream = false;
    hipEvent_t xstream_ev = nullptr;
    // AR loop of a small batch on two streams (hificar_ar_loop_ragged): the second stream, fork / join / schedule-upload events
    bool shared_chip = false;      // set while hificar_ar_loop runs two halves of a batch on two streams (launch_conv's tile choice)
    int ar_dual_min = 17, ar_dual_max = 62;  // HIFICAR_AR_DUAL_MIN / _MAX: the batch sizes the loop splits (max 0: never)
    hipStream_t ar_side = nullptr;
    hipEvent_t ar_ev[2] = {nullptr, nullptr};
    unsigned long long sched_up_seq = 0;  // schedule uploads so far (get_schedule): a schedule is uploaded on the stream that first needs it
    hipEvent_t done_ev = nullptr;  // recorded at the END of the last call on done_stream (calls that mark their end: the discriminators' backward)
    bool done_valid = false;
    hipStream_t done_stream = nullptr;
    // debug taps (hificar_debug_tap): name -> (destination, capacity in floats); scratch for pre-activation copies
    struct Tap {
        float* dst;
        size_t cap;
    };
    std::map<std::string, Tap> taps;
    float* tap_scratch = nullptr;
    size_t tap_scratch_elems = 0;
    // profiling (hificar_profile_begin/end)
    bool profiling = false;
    struct ProfRec {
        hipEvent_t e0, e1;
        std::string name;
        double flops, bytes;
    };
    std::vector<ProfRec> prof;
    hipStream_t prof_stream = nullptr;
};

// The library's environment switches — all of them, read once per handle (generator, GBlock generator, the discriminators' engine):
//   HIFICAR_PROFILE_DETAIL=1   rows of hificar_profile_end carry the layer name ("kernel|layer xN")
//   HIFICAR_LAUNCH_LOG=<path>  every kernel launch of the library is appended to <path> in enqueue order as "kernel|layer<TAB>flops<TAB>algorithmic bytes":
//                              joined with rocprofv3's per-dispatch rows by tools/pmc_by_layer.py (implies PROFILE_DETAIL)
//   HIFICAR_KSPLIT=0|1|2       split-K conv form: never / when estimated faster (default) / always.  0 makes every launch shape use one accumulation
//                              order, so results are bit-identical across batch compositions (tests/test_gpu_parity.py)
//   HIFICAR_PAIR=0             narrow stages layer by layer instead of the fused pair kernels;  HIFICAR_PAIR_SMALL=0: no 128-row pair tiles at C = 32
//   HIFICAR_AR_DUAL_MIN / _MAX the batch sizes hificar_ar_loop runs as two halves on two streams (default 17..62; MAX=0: never)
// hificar_disc.hip.inc adds HIFICAR_DISC_STREAMS=0 (sub-discriminators on the caller's stream: per-launch counters) and HIFICAR_COL2IM_VEC4=0.
static FILE* g_launch_log = nullptr;
static void read_env_switches(hificar_handle* h) {
    if (const char* e = getenv("HIFICAR_PROFILE_DETAIL")) h->profile_detail = atoi(e) != 0;
    if (const char* e = getenv("HIFICAR_LAUNCH_LOG")) {
        if (!g_launch_log && *e) g_launch_log = fopen(e, "a");
        if (g_launch_log) h->profile_detail = true;
    }
    if (const char* e = getenv("HIFICAR_KSPLIT")) h->ksplit = atoi(e);
    if (const char* e = getenv("HIFICAR_PAIR")) h->use_pair = atoi(e) != 0;
    if (const char* e = getenv("HIFICAR_PAIR_SMALL")) h->pair_small = atoi(e) != 0;
    if (const char* e = getenv("HIFICAR_AR_DUAL_MIN")) h->ar_dual_min = atoi(e);
    if (const char* e = getenv("HIFICAR_AR_DUAL_MAX")) h->ar_dual_max = atoi(e);
}

// RAII bracket around one kernel launch: the launch log, and an event before and after while profiling is on.
struct ProfScope {
    hificar_handle* h;
    hipStream_t s;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::string name;
    double flops, bytes;
    ProfScope(hificar_handle* h_, hipStream_t s_, const std::string& n, double f, double b) : h(h_), s(s_), name(n), flops(f), bytes(b) {
        if (g_launch_log) {
            fprintf(g_launch_log, "%s\t%.0f\t%.0f\n", n.c_str(), f, b);
            fflush(g_launch_log);
        }
        if (!h->profiling) return;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, s);
    }
    ~ProfScope() {
        if (!h->profiling) return;
        (void)hipEventRecord(e1, s);
        h->prof.push_back({e0, e1, name, flops, bytes});
        h->prof_stream = s;
    }
};

static size_t round_up_sz(size_t x, size_t m) { return (x + m - 1) / m * m; }
static int arena_add(hificar_handle* h, size_t min_bytes);
static int bucket_frames(int T);

static int round_up(int x, int m) { return (x + m - 1) / m * m; }
static int stage_channels(const hificar_config& c, int i) { return c.channels >> i; }  // channels // 2**i
static int stage_pad(const hificar_config& c, int i) { return round_up(stage_channels(c, i), 32); }  // row pitch of that stage's buffers

static int conv_index(const hificar_handle* h, int stage, int block, int dil) {
    int idx = 0;
    for (int b = 0; b < stage * h->cfg.n_blocks + block; ++b) idx += h->cfg.n_dilations[b % h->cfg.n_blocks];
    return idx + dil;
}

// Derive tap tables and blocking for one layer.
static int plan_layer(ConvLayer& L) {
    if (L.cin_pad < 32 || L.cin_pad % 16 != 0 || L.cin < 1 || L.cout < 1)
        return fail(HIFICAR_E_INVALID, "internal: %s: bad channel counts (cin %d, padded %d, cout %d)", L.name.c_str(), L.cin, L.cin_pad, L.cout);
    L.cout_pad = round_up(L.cout, 32);  // any width runs (hifigan.py:108-145 accepts any): the MFMA tiles are 32 channels wide
    if (!L.transposed) {
        if (L.K > kMaxTaps) return fail(HIFICAR_E_INVALID, "%s: kernel size %d > %d", L.name.c_str(), L.K, kMaxTaps);
        L.n_phase = 1;
        L.ntaps = L.K;
        for (int k = 0; k < L.K; ++k) {
            L.tap_off[0][k] = k * L.dilation - L.padding;
            L.tap_k[0][k] = k;
        }
    } else {
        const int s = L.stride, p = L.padding;
        if (s > kMaxPhase) return fail(HIFICAR_E_INVALID, "%s: upsample scale %d > %d", L.name.c_str(), s, kMaxPhase);
        L.n_phase = s;
        L.ntaps = (L.K + s - 1) / s;
        if (L.ntaps > kMaxTaps) return fail(HIFICAR_E_INVALID, "%s: too many taps", L.name.c_str());
        for (int r = 0; r < s; ++r) {
            // out[q*s + r] = sum_k x[q + (r + p - k)/s] * W[:, :, k] over k == (r + p) mod s
            const int k0 = (r + p) % s;
            for (int t = 0; t < L.ntaps; ++t) {
                const int k = k0 + t * s;
                // (a tap past the kernel's end has zero weights; its offset continues the progression the K loop steps through)
                L.tap_k[r][t] = k < L.K ? k : -1;
                L.tap_off[r][t] = (r + p - k) / s;  // exact: r + p - k is a multiple of s
            }
        }
    }
    for (int r = 0; r < L.n_phase; ++r)
        for (int t = 1; t < L.ntaps; ++t)
            if (L.tap_off[r][t] - L.tap_off[r][t - 1] != L.tap_off[0][1] - L.tap_off[0][0])
                return fail(HIFICAR_E_INVALID, "%s: tap offsets are not an arithmetic progression", L.name.c_str());
    L.off_min = 0;
    L.off_max = 0;
    for (int r = 0; r < L.n_phase; ++r)
        for (int t = 0; t < L.ntaps; ++t) {
            L.off_min = std::min(L.off_min, L.tap_off[r][t]);
            L.off_max = std::max(L.off_max, L.tap_off[r][t]);
        }
    L.cout_total = L.cout_pad * L.n_phase;
    // 16/32/64 channels per LDS item (XOR-swizzled rows), and at least two items per tile (out-buffer hand-off)
    L.chunk16 = (L.cin_pad % 64 == 0 && L.cin_pad >= 128) ? 64 : (L.cin_pad % 32 == 0 && L.cin_pad >= 64) ? 32 : 16;
    L.n_blocks32 = L.cout_total / 32;
    L.nb32_per_phase = L.cout_pad / 32;
    return HIFICAR_OK;
}

// ------------------------------------------------------------------------------------------------
// create / destroy
// ------------------------------------------------------------------------------------------------
extern "C" const char* hificar_last_error(void) { return g_err; }
extern "C" const char* hificar_version(void) { return "hificar 0.1 gfx950"; }

extern "C" int hificar_create(const hificar_config* cfg, hificar_handle** out) {
    if (!cfg || !out) return fail(HIFICAR_E_INVALID, "hificar_create: null argument");
    const hificar_config& c = *cfg;
    if (c.out_channels != 1) return fail(HIFICAR_E_INVALID, "out_channels=%d unsupported (PQMF multi-band output is out of scope)", c.out_channels);
    if (c.kernel_size % 2 != 1) return fail(HIFICAR_E_INVALID, "Kernel size must be odd number.");
    if (c.n_stages < 1 || c.n_stages > HIFICAR_MAX_STAGES) return fail(HIFICAR_E_INVALID, "n_stages=%d out of range", c.n_stages);
    if (c.n_blocks < 1 || c.n_blocks > HIFICAR_MAX_BLOCKS)
        return fail(HIFICAR_E_INVALID, "n_blocks=%d unsupported (1..%d residual blocks per stage)", c.n_blocks, HIFICAR_MAX_BLOCKS);
    if (c.use_ar && (c.ar_input > 1024 || c.ar_hidden > 512 || c.ar_output > 512 || c.ar_input < 1))
        return fail(HIFICAR_E_INVALID, "PastFCEncoder: ar_input must be <= 1024, ar_hidden / ar_output <= 512");
    if (c.use_ar && (c.ar_hidden % 4 != 0 || c.ar_output % 4 != 0))
        return fail(HIFICAR_E_INVALID, "PastFCEncoder hidden/output dims must be multiples of 4");
    if (!(c.lrelu_slope >= 0.f && c.lrelu_slope <= 1.f)) return fail(HIFICAR_E_INVALID, "negative_slope=%g outside [0, 1]", c.lrelu_slope);
    if (c.precision != HIFICAR_PREC_F32 && c.precision != HIFICAR_PREC_BF16X3)
        return fail(HIFICAR_E_INVALID, "unknown precision %d", c.precision);
    if ((c.channels >> c.n_stages) < 1) return fail(HIFICAR_E_INVALID, "channels=%d leaves no channels after %d halvings", c.channels, c.n_stages);

    hificar_handle* h = new hificar_handle();
    h->cfg = c;
    read_env_switches(h);
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            h->num_cus = prop.multiProcessorCount;
    }
    h->precision = c.precision;
    if (c.use_spk_id && (c.num_spk < 1 || c.spk_emb_size < 1 || c.in_channels > 1024)) {
        delete h;
        return fail(HIFICAR_E_INVALID, "use_spk_id needs num_spk and spk_emb_size (and in_channels <= 1024)");
    }
    if ((c.use_ph || c.use_ph_loss) && c.num_ph < 1) {
        delete h;
        return fail(HIFICAR_E_INVALID, "use_ph / use_ph_loss need num_ph");
    }
    if (c.use_ph && c.ph_emb_size < 1) {
        delete h;
        return fail(HIFICAR_E_INVALID, "use_ph needs ph_emb_size");
    }
    if (c.use_spk_id && c.use_ph) {
        delete h;
        // spk_fc maps to in_channels values but is added before the phoneme channels are appended: the shapes disagree in the reference
        return fail(HIFICAR_E_INVALID, "use_spk_id together with use_ph is ill-formed in the reference (hifigan.py:212-220)");
    }
    h->cf = c.in_channels - (c.use_ar ? c.ar_output : 0) - (c.use_ph ? c.ph_emb_size : 0);
    if (h->cf < 1) {
        delete h;
        return fail(HIFICAR_E_INVALID, "in_channels=%d leaves no feature channels", c.in_channels);
    }
    h->cin_pad = std::max(32, round_up(c.in_channels, 16));
    h->hop = 1;
    for (int i = 0; i < c.n_stages; ++i) h->hop *= c.upsample_scales[i];

    int rc = HIFICAR_OK;
    auto expect = [&](const std::string& n, std::vector<int64_t> s) { h->expected[n] = s; };

    ConvLayer& ic = h->input_conv;
    ic.name = "input_conv";
    ic.cin = c.in_channels;
    ic.cin_pad = h->cin_pad;
    ic.cout = c.channels;
    ic.K = c.kernel_size;
    ic.padding = (c.kernel_size - 1) / 2;
    rc = plan_layer(ic);
    expect("input_conv.weight", {c.channels, c.in_channels, c.kernel_size});
    expect("input_conv.bias", {c.channels});

    for (int i = 0; i < c.n_stages && rc == HIFICAR_OK; ++i) {
        const int s = c.upsample_scales[i], K = c.upsample_kernel_sizes[i];
        const int pad = s / 2 + s % 2, opad = s % 2;  // hifigan.py:82-103
        if (K - 2 * pad + opad != s) {
            rc = fail(HIFICAR_E_INVALID, "upsample stage %d: kernel %d / scale %d does not give L_out = scale*L_in", i, K, s);
            break;
        }
        ConvLayer u;
        u.name = "upsamples." + std::to_string(i) + ".1";
        u.transposed = true;
        u.cin = stage_channels(c, i);
        u.cin_pad = stage_pad(c, i);
        u.cout = stage_channels(c, i + 1);
        u.K = K;
        u.stride = s;
        u.padding = pad;
        rc = plan_layer(u);
        expect(u.name + ".weight", {u.cin, u.cout, K});
        expect(u.name + ".bias", {u.cout});
        h->ups.push_back(u);
        for (int j = 0; j < c.n_blocks && rc == HIFICAR_OK; ++j) {
            const int k = c.resblock_kernel_sizes[j];
            if (k % 2 != 1) {
                rc = fail(HIFICAR_E_INVALID, "Kernel size must be odd number.");
                break;
            }
            if (c.n_dilations[j] < 1 || c.n_dilations[j] > HIFICAR_MAX_DILATIONS) {
                rc = fail(HIFICAR_E_INVALID, "block %d: n_dilations out of range", j);
                break;
            }
            for (int d = 0; d < c.n_dilations[j] && rc == HIFICAR_OK; ++d) {
                const std::string base = "blocks." + std::to_string(i * c.n_blocks + j);
                ConvLayer c1;
                c1.name = base + ".convs1." + std::to_string(d) + ".1";
                c1.cin = c1.cout = u.cout;
                c1.cin_pad = stage_pad(c, i + 1);
                c1.K = k;
                c1.dilation = c.resblock_dilations[j][d];
                c1.padding = (k - 1) / 2 * c1.dilation;
                c1.has_bias = c.bias != 0;
                rc = plan_layer(c1);
                ConvLayer c2 = c1;
                c2.name = base + ".convs2." + std::to_string(d) + ".1";
                c2.dilation = 1;
                c2.padding = (k - 1) / 2;
                if (rc == HIFICAR_OK) rc = plan_layer(c2);
                // use_additional_convs = false (residual_block.py:151, 191-205): no convs2, a layer is x = x + conv1(LeakyReLU(x))
                for (const ConvLayer* l : {&c1, &c2}) {
                    if (l == &c2 && !c.use_additional_convs) continue;
                    expect(l->name + ".weight", {l->cout, l->cin, k});
                    if (l->has_bias) expect(l->name + ".bias", {l->cout});
                }
                h->convs1.push_back(c1);
                if (c.use_additional_convs) h->convs2.push_back(c2);
            }
        }
    }
    const int c_last = stage_channels(c, c.n_stages);
    h->c_last = c_last;
    expect("output_conv.1.weight", {1, c_last, c.kernel_size});
    expect("output_conv.1.bias", {1});
    if (c.use_ar) {
        int dims[6] = {c.ar_input, c.ar_hidden, c.ar_hidden, c.ar_hidden, c.ar_hidden, c.ar_output};
        for (int l = 0; l < 5; ++l) {
            expect("ar_model.model." + std::to_string(2 * l) + ".weight", {dims[l + 1], dims[l]});
            expect("ar_model.model." + std::to_string(2 * l) + ".bias", {dims[l + 1]});
        }
    }
    if (c.use_spk_id) {
        expect("spk_emb_mat.weight", {c.num_spk, c.spk_emb_size});
        expect("spk_fc.weight", {c.in_channels, c.spk_emb_size});
        expect("spk_fc.bias", {c.in_channels});
    }
    if (c.use_ph) expect("ph_emb_mat.weight", {c.num_ph, c.ph_emb_size});
    if (c.use_ph_loss) {
        if (c_last > 128) rc = fail(HIFICAR_E_INVALID, "use_ph_loss: the phoneme head handles up to 128 last-stage channels (got %d)", c_last);
        if (h->hop % 2 != 0) rc = fail(HIFICAR_E_INVALID, "use_ph_loss: prod(upsample_scales) must be even (hifigan.py:186)");
        expect("ph_fc.weight", {c.num_ph, c_last});
        expect("ph_fc.bias", {c.num_ph});
    }
    if (rc != HIFICAR_OK) {
        delete h;
        return rc;
    }
    *out = h;
    return HIFICAR_OK;
}

extern "C" void hificar_destroy(hificar_handle* h) {
    if (!h) return;
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->d_tab) (void)hipFree(h->d_tab);
    if (h->h_tab) (void)hipHostFree(h->h_tab);
    if (h->tap_scratch) (void)hipFree(h->tap_scratch);
    if (h->train_free) h->train_free(h);
    if (h->tab_copied) (void)hipEventDestroy(h->tab_copied);
    for (auto& a : h->arenas) {
        (void)hipFree(a.d);
        (void)hipHostFree(a.h);
    }
    if (h->xstream_ev) (void)hipEventDestroy(h->xstream_ev);
    if (h->ar_side) (void)hipStreamDestroy(h->ar_side);
    for (hipEvent_t e : h->ar_ev)
        if (e) (void)hipEventDestroy(e);
    if (h->done_ev) (void)hipEventDestroy(h->done_ev);
    for (auto& r : h->rslots) {
        if (r.d) (void)hipFree(r.d);
        if (r.h) (void)hipHostFree(r.h);
        if (r.ev) (void)hipEventDestroy(r.ev);
        if (r.up) (void)hipEventDestroy(r.up);
    }
    delete h;
}

extern "C" int hificar_set_weight(hificar_handle* h, const char* name, const float* data, const int64_t* shape, int ndim) {
    if (!h || !name || !data || !shape) return fail(HIFICAR_E_INVALID, "hificar_set_weight: null argument");
    if (h->finalized) return fail(HIFICAR_E_STATE, "hificar_set_weight(%s) after hificar_finalize", name);
    auto it = h->expected.find(name);
    if (it == h->expected.end()) return fail(HIFICAR_E_INVALID, "unexpected tensor name '%s' for this configuration", name);
    std::vector<int64_t> s(shape, shape + ndim);
    if (s != it->second) {
        std::string want, got;
        for (auto v : it->second) want += std::to_string(v) + ",";
        for (auto v : s) got += std::to_string(v) + ",";
        return fail(HIFICAR_E_INVALID, "size mismatch for %s: expected (%s) got (%s)", name, want.c_str(), got.c_str());
    }
    size_t n = 1;
    for (auto v : s) n *= (size_t)v;
    HostTensor t;
    t.shape = s;
    t.data.assign(data, data + n);
    h->tensors[name] = std::move(t);
    return HIFICAR_OK;
}

// ------------------------------------------------------------------------------------------------
// finalize: repack + upload
// ------------------------------------------------------------------------------------------------
static int upload(hificar_handle* h, const std::vector<float>& v, float** dptr) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(float)));
    h->allocs.push_back(p);
    HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    *dptr = static_cast<float*>(p);
    return HIFICAR_OK;
}

static inline uint16_t f32_to_bf16(float f) {  // round to nearest even (finite inputs)
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// bf16x3 path: weight fragments in MFMA lane order, [n_block32][chunk][tap][c16][hi|lo][lane][8]
static int pack_w16(hificar_handle* h, const ConvLayer& L, const HostTensor& W, int chunk, uint16_t** out) {
    const int nc16 = chunk / 16, nchunk = L.cin_pad / chunk;
    const size_t frag = 64 * 8;  // bf16 elements per fragment
    std::vector<uint16_t> w16(((size_t)L.n_blocks32 * nchunk * L.ntaps * nc16 * 2 + 4 * nc16) * frag, 0);
    for (int nb = 0; nb < L.n_blocks32; ++nb) {
        const int phase = nb / L.nb32_per_phase;
        const int co0 = (nb % L.nb32_per_phase) * 32;
        for (int c = 0; c < nchunk; ++c)
            for (int t = 0; t < L.ntaps; ++t) {
                const int k = L.tap_k[phase][t];
                if (k < 0) continue;
                for (int u = 0; u < nc16; ++u) {
                    uint16_t* hi = &w16[(((((size_t)nb * nchunk + c) * L.ntaps + t) * nc16 + u) * 2) * frag];
                    uint16_t* lo = hi + frag;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int g = lane >> 5, n = lane & 31, co = co0 + n;
                        if (co >= L.cout) continue;
                        for (int j = 0; j < 8; ++j) {
                            const int ci = c * chunk + u * 16 + 8 * g + j;
                            if (ci >= L.cin) continue;
                            const size_t src = L.transposed ? ((size_t)ci * L.cout + co) * L.K + k : ((size_t)co * L.cin + ci) * L.K + k;
                            const float v = W.data[src];
                            const uint16_t vh = f32_to_bf16(v);
                            hi[lane * 8 + j] = vh;
                            lo[lane * 8 + j] = f32_to_bf16(v - bf16_to_f32(vh));
                        }
                    }
                }
            }
    }
    void* dp = nullptr;
    HIP_TRY(hipMalloc(&dp, w16.size() * sizeof(uint16_t)));
    h->allocs.push_back(dp);
    HIP_TRY(hipMemcpy(dp, w16.data(), w16.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    *out = static_cast<uint16_t*>(dp);
    return HIFICAR_OK;
}

// exact-fp32 arithmetic: fp32 weight fragments, [n_block32][chunk][tap][c16][half][lane][4]: lane (n = lane & 31, g = lane >> 5)
// holds channels 16*c16 + 8*half + 4*g + {0..3} of output channel n (one v_mfma_f32_32x32x2_f32 step per element)
static int pack_w32(hificar_handle* h, const ConvLayer& L, const HostTensor& W, int chunk, float** out) {
    const int nc16 = chunk / 16, nchunk = L.cin_pad / chunk;
    const size_t frag = 64 * 4;  // floats per fragment
    std::vector<float> w32(((size_t)L.n_blocks32 * nchunk * L.ntaps * nc16 * 2 + 4 * nc16) * frag, 0.f);
    for (int nb = 0; nb < L.n_blocks32; ++nb) {
        const int phase = nb / L.nb32_per_phase;
        const int co0 = (nb % L.nb32_per_phase) * 32;
        for (int c = 0; c < nchunk; ++c)
            for (int t = 0; t < L.ntaps; ++t) {
                const int k = L.tap_k[phase][t];
                if (k < 0) continue;
                for (int u = 0; u < nc16; ++u)
                    for (int v = 0; v < 2; ++v) {
                        float* f = &w32[((((((size_t)nb * nchunk + c) * L.ntaps + t) * nc16 + u) * 2) + v) * frag];
                        for (int lane = 0; lane < 64; ++lane) {
                            const int g = lane >> 5, n = lane & 31, co = co0 + n;
                            if (co >= L.cout) continue;
                            for (int j = 0; j < 4; ++j) {
                                const int ci = c * chunk + u * 16 + 8 * v + 4 * g + j;
                                if (ci >= L.cin) continue;
                                const size_t src = L.transposed ? ((size_t)ci * L.cout + co) * L.K + k : ((size_t)co * L.cin + ci) * L.K + k;
                                f[lane * 4 + j] = W.data[src];
                            }
                        }
                    }
            }
    }
    return upload(h, w32, out);
}

static int pack_conv(hificar_handle* h, ConvLayer& L) {
    const HostTensor& W = h->tensors.at(L.name + ".weight");
    std::vector<float> bias((size_t)L.cout_total, 0.f);
    if (L.has_bias) {
        const HostTensor& Bv = h->tensors.at(L.name + ".bias");
        for (int r = 0; r < L.n_phase; ++r)
            for (int co = 0; co < L.cout; ++co) bias[(size_t)r * L.cout_pad + co] = Bv.data[co];
    }
    int rc = upload(h, bias, &L.d_bias);
    if (rc != HIFICAR_OK) return rc;

    if ((rc = pack_w16(h, L, W, L.chunk16, &L.d_w16)) != HIFICAR_OK) return rc;
    if ((rc = pack_w32(h, L, W, L.chunk16, &L.d_w32)) != HIFICAR_OK) return rc;
    if (!L.transposed && L.cin_pad == L.cin && (L.cin == 32 || L.cin == 64) && L.cout == L.cin) {
        if ((rc = pack_w16(h, L, W, L.cin_pad, &L.d_w16c)) != HIFICAR_OK) return rc;
        if ((rc = pack_w32(h, L, W, L.cin_pad, &L.d_w32c)) != HIFICAR_OK) return rc;
    }
    return HIFICAR_OK;
}

// the conv kernels' instantiation sets (hificar_conv_inst.hip): the one place the host code meets them
hipError_t hificar::conv_launch(const ConvShape& s, const void* params, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    bool handled = false;
    hipError_t e = hipSuccess;
#define HIFICAR_TRY_SET(n)                                                      \
    if (!handled) e = conv_inst_launch_##n(s, params, grid, lds_bytes, stream, &handled); \
    if (handled) return e;
    HIFICAR_TRY_SET(0) HIFICAR_TRY_SET(1) HIFICAR_TRY_SET(2) HIFICAR_TRY_SET(3) HIFICAR_TRY_SET(4)
    HIFICAR_TRY_SET(5) HIFICAR_TRY_SET(6) HIFICAR_TRY_SET(7) HIFICAR_TRY_SET(8) HIFICAR_TRY_SET(9)
#undef HIFICAR_TRY_SET
    return hipErrorInvalidValue;  // a shape that is not built
}

hipError_t hificar::conv_set_lds_attributes() {
    hipError_t e = hipSuccess;
#define HIFICAR_ATTR_SET(n) \
    if (e == hipSuccess) e = conv_inst_attrs_##n();
    HIFICAR_ATTR_SET(0) HIFICAR_ATTR_SET(1) HIFICAR_ATTR_SET(2) HIFICAR_ATTR_SET(3) HIFICAR_ATTR_SET(4)
    HIFICAR_ATTR_SET(5) HIFICAR_ATTR_SET(6) HIFICAR_ATTR_SET(7) HIFICAR_ATTR_SET(8) HIFICAR_ATTR_SET(9)
#undef HIFICAR_ATTR_SET
    return e;
}

// Device-side state every launch needs, whatever network the handle holds (the generator, or the discriminators' engine handle):
// the zero page of the LDS DMA, the kernels' dynamic-LDS attributes, the first schedule arena.
static int engine_setup(hificar_handle* h) {
    {
        void* z = nullptr;
        HIP_TRY(hipMalloc(&z, 256));
        h->allocs.push_back(z);
        HIP_TRY(hipMemset(z, 0, 256));
        h->d_zeros = static_cast<char*>(z);
    }
    HIP_TRY(conv_set_lds_attributes());
    if (h->arenas.empty()) {
        int rca = arena_add(h, 0);
        if (rca != HIFICAR_OK) return rca;
    }
    return HIFICAR_OK;
}

extern "C" int hificar_finalize(hificar_handle* h) {
    if (!h) return fail(HIFICAR_E_INVALID, "hificar_finalize: null handle");
    if (h->finalized) return HIFICAR_OK;
    for (auto& kv : h->expected)
        if (!h->tensors.count(kv.first)) return fail(HIFICAR_E_STATE, "Missing key(s) in state_dict: \"%s\"", kv.first.c_str());
    int rc;
    if ((rc = pack_conv(h, h->input_conv)) != HIFICAR_OK) return rc;
    for (auto& l : h->ups)
        if ((rc = pack_conv(h, l)) != HIFICAR_OK) return rc;
    for (auto& l : h->convs1)
        if ((rc = pack_conv(h, l)) != HIFICAR_OK) return rc;
    for (auto& l : h->convs2)
        if ((rc = pack_conv(h, l)) != HIFICAR_OK) return rc;
    for (auto& g : h->gb)
        for (ConvLayer* l : {&g.c1a, &g.c1b, &g.res, &g.c2a, &g.c2b})
            if ((rc = pack_conv(h, *l)) != HIFICAR_OK) return rc;
    {   // output conv weight (1, C, K) -> [k][C]
        const HostTensor& W = h->tensors.at("output_conv.1.weight");
        const int C = (int)W.shape[1], K = (int)W.shape[2], Cp = round_up(C, 32);  // [k][padded channels]
        std::vector<float> w((size_t)Cp * K, 0.f);
        for (int ch = 0; ch < C; ++ch)
            for (int k = 0; k < K; ++k) w[(size_t)k * Cp + ch] = W.data[(size_t)ch * K + k];
        if ((rc = upload(h, w, &h->d_out_w)) != HIFICAR_OK) return rc;
        h->out_bias = h->tensors.at("output_conv.1.bias").data[0];
    }
    if (h->cfg.use_ar) {
        for (int l = 0; l < 5; ++l) {
            const HostTensor& W = h->tensors.at("ar_model.model." + std::to_string(2 * l) + ".weight");
            const int dout = (int)W.shape[0], din = (int)W.shape[1];
            std::vector<float> wt((size_t)din * dout);
            for (int o = 0; o < dout; ++o)
                for (int i = 0; i < din; ++i) wt[(size_t)i * dout + o] = W.data[(size_t)o * din + i];
            if ((rc = upload(h, wt, &h->d_mlp_w[l])) != HIFICAR_OK) return rc;
            if ((rc = upload(h, h->tensors.at("ar_model.model." + std::to_string(2 * l) + ".bias").data, &h->d_mlp_b[l])) != HIFICAR_OK) return rc;
        }
    }
    if (h->cfg.use_spk_id) {
        if ((rc = upload(h, h->tensors.at("spk_emb_mat.weight").data, &h->d_spk_emb)) != HIFICAR_OK) return rc;
        if ((rc = upload(h, h->tensors.at("spk_fc.weight").data, &h->d_spk_w)) != HIFICAR_OK) return rc;
        if ((rc = upload(h, h->tensors.at("spk_fc.bias").data, &h->d_spk_b)) != HIFICAR_OK) return rc;
    }
    if (h->cfg.use_ph && (rc = upload(h, h->tensors.at("ph_emb_mat.weight").data, &h->d_ph_emb)) != HIFICAR_OK) return rc;
    if (h->cfg.use_ph_loss) {
        if ((rc = upload(h, h->tensors.at("ph_fc.weight").data, &h->d_phfc_w)) != HIFICAR_OK) return rc;
        if ((rc = upload(h, h->tensors.at("ph_fc.bias").data, &h->d_phfc_b)) != HIFICAR_OK) return rc;
    }
    if ((rc = engine_setup(h)) != HIFICAR_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());
    h->tensors.clear();  // host copies no longer needed
    h->finalized = true;
    return HIFICAR_OK;
}

extern "C" int hificar_set_precision(hificar_handle* h, int precision) {
    if (!h) return fail(HIFICAR_E_INVALID, "null handle");
    if (h->arch == 1 && precision != HIFICAR_PREC_F32)
        return fail(HIFICAR_E_INVALID, "GBlockGenerator runs in the exact-fp32 arithmetic only (its 1 x 1 residual conv reads raw, un-activated rows)");
    if (precision == HIFICAR_PREC_F32 || precision == HIFICAR_PREC_BF16X3) {
        h->precision = precision;
        return HIFICAR_OK;
    }
    return fail(HIFICAR_E_INVALID, "unknown precision %d", precision);
}

// ------------------------------------------------------------------------------------------------
// workspace plan
// ------------------------------------------------------------------------------------------------
constexpr int kMaxBlk = HIFICAR_MAX_BLOCKS;  // residual blocks per stage (hifigan.py:134-145: one per resblock_kernel_sizes entry)

struct Workspace {
    // "activated rows" = LeakyReLU(x) as split rows (bf16x3) or plain fp32 rows (exact fp32): 4 bytes per element either way
    float* xin;    // (B, T, cin_pad) assembled input rows (no activation in front of the input conv)
    float* h0;     // input conv output, activated rows
    float* u;      // upsample output, fp32 (residual of the first ResBlock layer)
    float* x[kMaxBlk];   // per-branch residual stream, fp32
    float* xt[kMaxBlk];  // conv1 output, activated rows ([0] also holds the activated MRF mean for the next upsampler)
    char* u_s;           // activated rows of u
    char* x_s[kMaxBlk];  // activated rows of x_j
    size_t bytes;
};

static size_t stage_elems(const hificar_handle* h, int B, int T) {
    size_t mx = 0, L = (size_t)T;
    if (h->arch == 1) {  // GBlockGenerator: the input conv's output and every GBlock's output live in stage buffers
        mx = L * (size_t)stage_pad(h->cfg, 0);
        for (const GBlockLayers& g : h->gb) {
            L *= (size_t)g.scale;
            mx = std::max(mx, L * (size_t)std::max(g.c1a.cin_pad, g.c1a.cout_pad));
        }
        return mx * (size_t)B;
    }
    for (int i = 0; i < h->cfg.n_stages; ++i) {
        L *= h->cfg.upsample_scales[i];
        mx = std::max(mx, L * (size_t)stage_pad(h->cfg, i + 1));
    }
    return mx * (size_t)B;
}

// One layout for both arithmetics (set_precision may switch a live handle).
static Workspace plan_workspace(const hificar_handle* h, int B, int T, void* base) {
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t elems) {
        float* p = base ? reinterpret_cast<float*>(static_cast<char*>(base) + off) : nullptr;
        off += round_up_sz(elems * sizeof(float), 1024);
        return p;
    };
    w.xin = take((size_t)B * T * h->cin_pad);
    w.h0 = take((size_t)B * T * stage_pad(h->cfg, 0));
    const size_t se = stage_elems(h, B, T);
    w.u = take(se);
    const int nbw = std::max(3, h->cfg.n_blocks);  // (three sets at least: the GBlock engine and the MRF-mean ping-pong use them by index)
    for (int j = 0; j < kMaxBlk; ++j) w.x[j] = j < nbw ? take(se) : nullptr;
    for (int j = 0; j < kMaxBlk; ++j) w.xt[j] = j < nbw ? take(se) : nullptr;
    w.u_s = reinterpret_cast<char*>(take(se));
    for (int j = 0; j < kMaxBlk; ++j) w.x_s[j] = j < nbw ? reinterpret_cast<char*>(take(se)) : nullptr;
    w.bytes = off;
    return w;
}

extern "C" size_t hificar_workspace_bytes(const hificar_handle* h, int B, int T) {
    if (!h || B < 1 || T < 1) return 0;
    const int Tb = bucket_frames(T);
    size_t n = plan_workspace(h, B, Tb, nullptr).bytes;
    if (B >= 2)  // room for the two halves of the batch side by side (hificar_ar_loop on two streams; buffer sizes round up to 1 KB)
        n = std::max(n, plan_workspace(h, (B + 1) / 2, Tb, nullptr).bytes + plan_workspace(h, B / 2, Tb, nullptr).bytes);
    return n;
}

extern "C" double hificar_macs(const hificar_handle* h, int B, int T) {
    if (!h) return 0.0;
    const hificar_config& c = h->cfg;
    double m = (double)T * c.in_channels * c.channels * c.kernel_size;
    double L = T;
    for (const GBlockLayers& g : h->gb) {  // arch 1: four k-tap convs and the 1 x 1 residual conv per output row
        L *= g.scale;
        m += L * ((double)g.cin * g.cout * (g.c1a.K + 1) + 3.0 * g.cout * g.cout * g.c1a.K);
    }
    if (h->arch == 1) {
        m += L * h->c_last * c.kernel_size;
        if (c.use_ar) m += (double)c.ar_input * c.ar_hidden + 3.0 * c.ar_hidden * c.ar_hidden + (double)c.ar_hidden * c.ar_output;
        return m * B;
    }
    for (int i = 0; i < c.n_stages; ++i) {
        const double cin = stage_channels(c, i), cout = stage_channels(c, i + 1);
        m += L * cin * cout * c.upsample_kernel_sizes[i];
        L *= c.upsample_scales[i];
        for (int j = 0; j < c.n_blocks; ++j) m += L * cout * cout * c.resblock_kernel_sizes[j] * (c.use_additional_convs ? 2.0 : 1.0) * c.n_dilations[j];
    }
    m += L * stage_channels(c, c.n_stages) * c.kernel_size;
    if (c.use_ar) m += (double)c.ar_input * c.ar_hidden + 3.0 * c.ar_hidden * c.ar_hidden + (double)c.ar_hidden * c.ar_output;
    return m * B;
}

// ------------------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------------------
// Ragged batch context of one forward: utterance b has seq_len[b] frames (device array, null = all equal); the forward
// covers frames [f0, f0 + frames) of every utterance.
// Training tape: every activated conv input of one forward gets its own buffer (the backward pass reads them: wgrad operands and
// LeakyReLU' masks); nothing is overwritten.  Filled by plan_tape() from a caller-provided tape buffer.
struct Tape {
    float* xin = nullptr;                                              // (B, T, cin_pad) assembled input rows
    char* h0_s = nullptr;                                              // activated input-conv output
    char* upin_s[HIFICAR_MAX_STAGES] = {};                             // activated input of upsample i (i = 0: h0_s)
    char* u_s[HIFICAR_MAX_STAGES] = {};                                // activated upsample output
    char* xt_s[HIFICAR_MAX_STAGES][kMaxBlk][HIFICAR_MAX_DILATIONS] = {};     // activated conv1 output (= conv2 input)
    char* x_s[HIFICAR_MAX_STAGES][kMaxBlk][HIFICAR_MAX_DILATIONS] = {};      // activated residual stream after pair d (= next conv1 input)
    float* mlp = nullptr;                                              // (B, 5, 1024) PastFCEncoder layer inputs
    float* fin[kMaxBlk] = {};                                          // fp32 ResBlock outputs of the LAST stage (output conv backward)
    // GBlockGenerator (arch 1), per GBlock: the raw and the ReLU'd input at the block's OUTPUT rate (nearest-upsampled copies: the weight
    // gradients of res1 / conv1's first conv contract over them), and the ReLU'd inputs of the other three convs
    char* g_xu[HIFICAR_MAX_GBLOCKS] = {};
    char* g_a1u[HIFICAR_MAX_GBLOCKS] = {};
    char* g_a2[HIFICAR_MAX_GBLOCKS] = {};
    char* g_a3[HIFICAR_MAX_GBLOCKS] = {};
    char* g_a4[HIFICAR_MAX_GBLOCKS] = {};
    size_t bytes = 0;
};

// Conditioning inputs / extra output of one forward (hificar_forward_cond)
struct Cond {
    const int32_t* spk_id = nullptr;
    const int32_t* ph = nullptr;
    int ph_stride = 0;
    float* ph_out = nullptr;
    int ph_out_T = 0;
};

struct Ragged {
    const int32_t* seq_len = nullptr;
    int const_len = -1;  // >= 0 (and seq_len null): every utterance has this many frames, fewer than the launch covers (bucketed lengths)
    int f0 = 0;
    int frames = 0;
};

// Launch geometry of a non-AR forward is rounded up to a bucket of frames (the extra frames are masked exactly like the tail of a
// ragged batch: bit-identical results), so that a dataset of many distinct utterance lengths shares launch shapes / schedules.
static int bucket_frames(int T) { return T <= 32 ? T : round_up(T, 32); }

static void fill_params(ConvParams& p, const ConvLayer& L, int rows, int TM, const float* res, float* y, const Ragged& rg) {
    p.seq_len = rg.seq_len;
    p.len_const = rg.seq_len ? -1 : rg.const_len;
    p.len_f0 = rg.f0;
    p.len_max = rg.frames;
    p.len_mul = rg.frames > 0 ? rows / rg.frames : 1;
    p.w16 = reinterpret_cast<const bf16x8*>(L.d_w16);
    p.n_blocks32 = L.n_blocks32;
    p.nb32_per_phase = L.nb32_per_phase;
    p.bias = L.d_bias;
    p.res = res;
    p.y = y;
    p.L = rows;
    p.tiles_per_seq = (rows + TM - 1) / TM;
    p.cin = L.cin_pad;
    p.cout_total = L.cout_total;
    p.ntaps = L.ntaps;
    p.off_min = L.off_min;
    p.halo = L.off_max - L.off_min;
    p.tap_step = L.ntaps > 1 ? L.tap_off[0][1] - L.tap_off[0][0] : 0;
    for (int r = 0; r < kMaxPhase; ++r) p.tap_off0[r] = r < L.n_phase ? L.tap_off[r][0] : 0;
}

static constexpr size_t kArenaBytes = 16u << 20;
static constexpr size_t kMaxArenas = 8;

static int arena_add(hificar_handle* h, size_t min_bytes) {
    hificar_handle::Arena a;
    a.cap = std::max(kArenaBytes, round_up_sz(min_bytes, 4096));
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, a.cap));
    a.d = static_cast<char*>(p);
    hipError_t e = hipHostMalloc(&p, a.cap, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipFree(a.d);
        return fail(HIFICAR_E_HIP, "hipHostMalloc(schedule arena) failed: %s", hipGetErrorString(e));
    }
    a.h = static_cast<char*>(p);
    h->arenas.push_back(a);
    return HIFICAR_OK;
}

// `bytes` of arena space: device pointer + the pinned host mirror to fill before the upload
static int arena_take(hificar_handle* h, size_t bytes, char** d, char** hm) {
    bytes = round_up_sz(bytes, 256);
    if (h->arenas.empty() || h->arenas.back().used + bytes > h->arenas.back().cap) {
        if (h->arenas.size() >= kMaxArenas) {
            // a very large number of distinct launch shapes: recycle.  Kernels in flight may still read old schedules, so drain
            // the device first (rare, off the hot path; hificar_forward buckets non-AR lengths so that keys repeat)
            HIP_TRY(hipDeviceSynchronize());
            h->scheds.clear();
            for (auto& a : h->arenas) a.used = 0;
            std::sort(h->arenas.begin(), h->arenas.end(), [](const hificar_handle::Arena& x, const hificar_handle::Arena& y) { return x.cap < y.cap; });
            if (bytes > h->arenas.back().cap) return fail(HIFICAR_E_INVALID, "tile schedule of %zu bytes exceeds the arena", bytes);
        } else {
            int rc = arena_add(h, bytes);  // allocates: only when the pre-allocated arena is full
            if (rc != HIFICAR_OK) return rc;
        }
    }
    hificar_handle::Arena& a = h->arenas.back();
    *d = a.d + a.used;
    *hm = a.h + a.used;
    a.used += bytes;
    return HIFICAR_OK;
}

// Longest-processing-time-first assignment of `costs.size()` tiles to G workgroups; each workgroup's list is then
// ordered light -> heavy (the kernel walks it in that order).  Built once per launch shape, uploaded asynchronously on the
// launch stream (the launch that follows is ordered behind the copy) and cached.
static int put_schedule(hificar_handle* h, const std::string& key, const std::vector<std::vector<int>>& lists, int n, hipStream_t stream,
                        const int** d_start, const int** d_tiles);

static int get_schedule(hificar_handle* h, const std::string& key, const std::vector<double>& costs, int G, hipStream_t stream,
                        const int** d_start, const int** d_tiles) {
    auto it = h->scheds.find(key);
    if (it == h->scheds.end()) {
        const int n = (int)costs.size();
        std::vector<int> order(n);
        for (int i = 0; i < n; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return costs[a] > costs[b]; });
        std::vector<std::vector<int>> lists(G);
        // min-heap of (load, workgroup)
        std::vector<std::pair<double, int>> heap;
        for (int w = 0; w < G; ++w) heap.push_back({0.0, w});
        auto cmp = [](const std::pair<double, int>& a, const std::pair<double, int>& b) {
            return a.first > b.first || (a.first == b.first && a.second > b.second);
        };
        std::make_heap(heap.begin(), heap.end(), cmp);
        for (int t : order) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            auto& top = heap.back();
            lists[top.second].push_back(t);
            top.first += costs[t];
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
        for (auto& l : lists) std::reverse(l.begin(), l.end());  // heavy-first insertion order -> light first
        return put_schedule(h, key, lists, n, stream, d_start, d_tiles);
    }
    *d_start = it->second.d_start;
    *d_tiles = it->second.d_tiles;
    return HIFICAR_OK;
}

// An explicit tile list per workgroup (walked in the order given), uploaded asynchronously on the launch stream and cached under `key`.
static int put_schedule(hificar_handle* h, const std::string& key, const std::vector<std::vector<int>>& lists, int n, hipStream_t stream,
                        const int** d_start, const int** d_tiles) {
    auto it = h->scheds.find(key);
    if (it == h->scheds.end()) {
        const int G = (int)lists.size();
        const size_t n_start = (size_t)G + 1;
        const size_t bytes = (round_up_sz(n_start, 4) + (size_t)std::max(n, 1)) * sizeof(int);
        char *dp = nullptr, *hp = nullptr;
        int rc = arena_take(h, bytes, &dp, &hp);
        if (rc != HIFICAR_OK) return rc;
        int* start = reinterpret_cast<int*>(hp);
        int* tiles = start + round_up_sz(n_start, 4);
        int pos = 0;
        for (int w = 0; w < G; ++w) {
            start[w] = pos;
            for (int t : lists[w]) tiles[pos++] = t;
        }
        start[G] = pos;
        HIP_TRY(hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, stream));  // pinned source, never rewritten while cached
        ++h->sched_up_seq;
        hificar_handle::Sched sc;
        sc.d_start = reinterpret_cast<int*>(dp);
        sc.d_tiles = sc.d_start + round_up_sz(n_start, 4);
        it = h->scheds.emplace(key, sc).first;
    }
    *d_start = it->second.d_start;
    *d_tiles = it->second.d_tiles;
    return HIFICAR_OK;
}

// Order this call's work behind the previous call's when the caller changed streams (shared handle state: schedules, step
// table, workspace).  Same stream: nothing to do.
static int enter_stream(hificar_handle* h, hipStream_t stream) {
    if (h->have_last_stream && h->last_stream != stream) {
        if (h->done_valid && h->done_stream == h->last_stream) {  // the previous call marked its own end: wait for that, not for later work
            HIP_TRY(hipStreamWaitEvent(stream, h->done_ev, 0));
        } else {
            if (!h->xstream_ev) HIP_TRY(hipEventCreateWithFlags(&h->xstream_ev, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(h->xstream_ev, h->last_stream));
            HIP_TRY(hipStreamWaitEvent(stream, h->xstream_ev, 0));
        }
    }
    h->done_valid = false;
    h->last_stream = stream;
    h->have_last_stream = true;
    return HIFICAR_OK;
}

struct TileCfg {
    int MI, WM, WN;
    int KS;  // 1: each MFMA wave owns a 32-channel block of the tile; 4: the four MFMA waves split the K loop of ONE block (WM = WN = 1)
    int NB;  // channel blocks per MFMA wave (conv_ws_body's register blocking, bf16x3 only): tile = WM*MI*32 rows x WN*NB*32 channels
};
static size_t out_buf_bytes(const TileCfg& t) { return (size_t)t.KS * (t.WM * t.MI * 32) * (t.WN * t.NB * 32 + 4) * sizeof(float); }
// preference order: ties keep the earlier entry (taller wave tiles re-read fewer weights per MFMA)
constexpr int kNumTileCfgs = 15;
static const TileCfg kTileCfgs[kNumTileCfgs] = {{4, 1, 4, 1, 1}, {4, 2, 2, 1, 1}, {4, 4, 1, 1, 1}, {2, 2, 2, 1, 2}, {2, 4, 1, 1, 2},
                                                {2, 1, 4, 1, 2}, {2, 1, 4, 1, 1}, {2, 2, 2, 1, 1}, {2, 4, 1, 1, 1},
                                                {1, 1, 4, 1, 1}, {1, 2, 2, 1, 1}, {1, 4, 1, 1, 1}, {4, 1, 1, 4, 1}, {2, 1, 1, 4, 1}, {1, 1, 1, 4, 1}};

// One conv launch on activated rows (either arithmetic).  Per branch: xs (activated input) -> y (fp32, nullable) and/or ys (split copy of
// LeakyReLU(out, slope_out), nullable), + optional fp32 residual.
struct ConvIO {
    const char* xs;
    const float* res;
    float* y;
    char* ys;
    const float* mask_src = nullptr;  // backward launches: LeakyReLU'(mask_src) scales the result before res is added
    float mask_slope = 0.f;
    long long x_seq_bytes = 0;        // input row addressing (ConvParams::x_seq_bytes / x_row_bytes); 0: packed rows
    int x_row_bytes = 0;
    int x_rows = 0;                   // input rows per sequence when they differ from the launch's rows (ConvParams::x_rows)
    int x_up = 0;                     // nearest-neighbour upsampling of the input rows while staging (ConvParams::x_up); the input then has rows / x_up rows
    float x_slope = -1.f;             // >= 0: xs holds PRE-activation fp32 rows, LeakyReLU(x_slope) is applied while staging (ConvParams::act_in; exact fp32 only)
    const float* x_more[3] = {nullptr, nullptr, nullptr};  // ... and the input is the mean of xs and these further streams (x_n in all), summed in this order
    int x_n = 1;
};

// Replicas of every branch at regular strides (the groups of a grouped conv): see MultiConvParams::zrep
struct ConvRep {
    int n = 1;
    long long zs_x = 0, zs_w = 0, zs_y = 0, zs_b = 0;
};

static int launch_conv(hificar_handle* h, const ConvLayer* const* layers, int nbr, int nseq, int rows, const ConvIO* io,
                              float slope_out, const Ragged& rg, hipStream_t stream, const ConvRep& zr = ConvRep()) {
    const ConvLayer& L0 = *layers[0];
    const bool f32 = h->precision == HIFICAR_PREC_F32;  // rows are plain fp32 LeakyReLU(x) instead of split rows
    int halo_all = 0;
    for (int b = 0; b < nbr; ++b) halo_all = std::max(halo_all, layers[b]->off_max - layers[b]->off_min);
    // Tile shape: simulate the kernel's static tile walk (workgroup w takes tiles w, w+G, ...; branch-major order) and
    // take the shape with the smallest makespan.  A tile costs its MFMA issue cycles (all four MFMA waves run in
    // lock step: 3*MI MFMAs of 32 cycles per 16-channel K slab) plus a fixed per-tile and per-item overhead.
    TileCfg tc = {1, 4, 1, 1, 1};
    double best = 1e300;
    int best_ti = 11;  // {1, 4, 1, 1, 1}
    // The dense exact-fp32 launches are direct-output (conv_f32do_kernel): tiles stored straight from the MFMA waves' accumulators, no LDS out-buffer
    // (its bytes are free for taller tiles / wider halos below).  The register-blocked (NB = 2) wave tiles exist in bf16x3 only, where the cost model
    // may prefer them (+1.3 %; in exact fp32 they measured -1.0 %: profiles/r04_nb_register_blocking.txt).
    const int nsteps_min = [&] {
        int m = 1 << 30;
        for (int b = 0; b < nbr; ++b) m = std::min(m, layers[b]->ntaps * (L0.chunk16 / 16));
        return m;
    }();
    std::string pick_key;
    pick_key.reserve(160);
    pick_key += f32 ? 'f' : 'c';
    pick_key += h->train ? 't' : 'i';
    if (h->shared_chip) pick_key += 's';  // (launches that share the chip with another stream's: hificar_ar_loop on two streams)
    for (int b = 0; b < nbr; ++b) {
        pick_key += '|';
        pick_key += layers[b]->name;
    }
    pick_key += '|' + std::to_string(nseq) + 'x' + std::to_string(rows) + 'z' + std::to_string(zr.n);
    const auto cached_pick = h->tile_picks.find(pick_key);
    if (cached_pick != h->tile_picks.end()) tc = kTileCfgs[cached_pick->second];
    else
    for (int ti = 0; ti < kNumTileCfgs; ++ti) {
        const TileCfg& t = kTileCfgs[ti];
        const int TM = t.WM * t.MI * 32;
        const int chunk = L0.chunk16, RB = chunk * 4;
        if (t.NB == 2 && (f32 || L0.chunk16 == 16)) continue;  // (not instantiated)
        const size_t obuf = (f32 && t.KS == 1) ? 0 : out_buf_bytes(t);
        if (2 * round_up_sz((size_t)(TM + halo_all) * RB, 1024) + obuf > 160 * 1024) continue;
        if (t.NB == 2) {  // a wave's two channel blocks share the activation fragments: same phase of a polyphase (transposed) conv
            bool ok = L0.n_blocks32 >= 2;
            for (int b = 0; b < nbr; ++b) ok = ok && (layers[b]->n_phase == 1 || layers[b]->nb32_per_phase % 2 == 0);
            if (!ok) continue;
        }
        if (t.KS == 4 && (h->ksplit == 0 || nsteps_min < 2)) continue;
        if (t.KS == 1 && h->ksplit == 2 && nsteps_min >= 2) continue;
        const long long tiles_per_branch = (long long)nseq * ((rows + TM - 1) / TM) * ((L0.n_blocks32 + t.WN * t.NB - 1) / (t.WN * t.NB));
        const long long total = tiles_per_branch * nbr * zr.n;
        const int G = (int)std::min<long long>(total, h->num_cus);
        const int nchunks = L0.cin_pad / chunk;
        // LPT makespan estimate: max(heaviest tile, total / G), plus one light tile when the count does not divide
        double total_cost = 0.0, heaviest = 0.0, lightest = 1e300, cb[3] = {0.0, 0.0, 0.0};
        for (int b = 0; b < nbr; ++b) {
            // MFMA issue cycles per 16-channel slab and 32x32 accumulator: 3 x 32 (bf16x3, ~75 % sustained) or 8 x 64 (fp32)
            const double slab = f32 ? 8 * 64.0 : 3 * 32 / 0.75;
            double c = (double)layers[b]->ntaps * (L0.cin_pad / 16) * slab * t.MI * t.NB + 2500.0 + 400.0 * nchunks;
            if (t.KS == 4) {
                // each wave runs ceil(steps / 4) of an item's (tap, slab) steps; exposed weight latency at every tile start, the partial
                // sums' extra pass, and four times the staging per output
                const int steps = layers[b]->ntaps * (chunk / 16);
                c = (double)((steps + 3) / 4) * nchunks * slab * t.MI + 4000.0 + 600.0 * nchunks;
            }
            total_cost += c * tiles_per_branch * zr.n;
            heaviest = std::max(heaviest, c);
            lightest = std::min(lightest, c);
            cb[b] = c;
        }
        double worst = std::max(heaviest, total_cost / G);
        if (h->use_lpt && !h->shared_chip && total > G && total <= 4096) {
            // the makespan of the assignment the kernel will actually walk (get_schedule: longest tile first onto the least loaded workgroup).
            // Round 4: the closed form below charged "+ half a light tile" whenever the tile count is not a multiple of the workgroups — 384
            // tiles of weight 12 : 8 : 4 on 256 workgroups balance exactly (12 | 8 + 4), and the 128-row tile it ruled out at C = 256 is 1.1 %
            // faster end to end than the 64-row one it picked.
            std::sort(cb, cb + nbr, [](double x, double y) { return x > y; });
            std::priority_queue<double, std::vector<double>, std::greater<double>> load;
            for (int w = 0; w < G; ++w) load.push(0.0);
            worst = 0.0;
            for (int b = 0; b < nbr; ++b)
                for (long long i = 0; i < tiles_per_branch * zr.n; ++i) {
                    const double v = load.top() + cb[b];
                    load.pop();
                    load.push(v);
                    worst = std::max(worst, v);
                }
        } else if (total % G != 0) {
            worst = std::max(worst, total_cost / G + 0.5 * lightest);
        }
        // An engine whose launches overlap with others on side streams (the discriminators): a launch's own makespan — a nearly empty
        // last round, a chip half filled by tall tiles — is filled by the neighbours' workgroups, so what counts is the workgroup-time the
        // shape costs, i.e. its efficiency per tile.  (Measured: discriminator step 20.6 -> 19.9 ms, generator-side pass 10.7 -> 10.0 ms;
        // launches with fewer tiles than pick_throughput stay latency-driven.)
        if (h->pick_throughput > 0 && total >= h->pick_throughput) worst = total_cost / h->num_cus;
        // shorter wave tiles re-read the weight stream more often per MFMA (MI = 2 measured ~10 % slower per flop)
        // operand loads per MFMA: 2 (MI + NB) 16-byte loads per slab step against (8 | 3) MI NB MFMAs
        if (t.NB == 2) worst *= f32 ? (t.MI == 4 ? 0.95 : t.MI == 2 ? 0.975 : 1.02) : (t.MI == 2 ? 0.93 : 1.10);
        else if (t.MI < 4) worst *= f32 ? (t.MI == 2 ? 1.02 : h->mi1_penalty) : (t.MI == 2 ? 1.10 : 1.25);
        if (t.KS == 4) worst *= 1.05;  // near-ties go to the dense form
        if (worst < best * 0.98) {  // near-ties keep the earlier (taller) shape
            best = worst;
            tc = t;
            best_ti = ti;
        }
    }
    if (cached_pick == h->tile_picks.end()) {
        if (h->tile_picks.size() > 20000) h->tile_picks.clear();  // (a very large number of distinct launch shapes: start over)
        h->tile_picks.emplace(pick_key, best_ti);
    }
    const int TM = tc.WM * tc.MI * 32;
    const int chunk_sel = L0.chunk16, RB = chunk_sel * 4;
    const int nc16 = chunk_sel / 16;
    MultiConvParams mp;
    memset(&mp, 0, sizeof(mp));
    double flops = 0.0, bytes = 0.0;
    for (int b = 0; b < nbr; ++b) {
        const ConvLayer& Lb = *layers[b];
        if (Lb.n_blocks32 != L0.n_blocks32 || Lb.chunk16 != L0.chunk16 || Lb.cin_pad != L0.cin_pad)
            return fail(HIFICAR_E_INVALID, "internal: branch shape mismatch");
        fill_params(mp.p[b], Lb, rows, TM, io[b].res, io[b].y, rg);
        mp.p[b].xs = io[b].xs;
        mp.p[b].ys = io[b].ys;
        if (io[b].x_slope >= 0.f) {
            if (!f32) return fail(HIFICAR_E_INVALID, "internal: pre-activation input rows of %s outside the exact-fp32 layer-by-layer launches", Lb.name.c_str());
            if (io[b].x_n < 1 || io[b].x_n > 4) return fail(HIFICAR_E_INVALID, "internal: %d input streams of %s", io[b].x_n, Lb.name.c_str());
            mp.p[b].act_in = io[b].x_n;
            for (int q = 0; q + 1 < io[b].x_n; ++q) mp.p[b].xs_more[q] = reinterpret_cast<const char*>(io[b].x_more[q]);
            mp.p[b].slope_in = io[b].x_slope;
        }
        mp.p[b].mask_src = io[b].mask_src;
        mp.p[b].mask_slope = io[b].mask_slope;
        mp.p[b].x_seq_bytes = io[b].x_seq_bytes;
        mp.p[b].x_row_bytes = io[b].x_row_bytes;
        mp.p[b].x_rows = io[b].x_rows;
        if (io[b].x_up > 1) {
            if (rows % io[b].x_up != 0 || io[b].x_seq_bytes != 0 || io[b].x_row_bytes != 0)
                return fail(HIFICAR_E_INVALID, "internal: upsampled input rows of %s", Lb.name.c_str());
            mp.p[b].x_up = io[b].x_up;
            mp.p[b].x_up_rcp = (unsigned)(0x100000000ull / (unsigned)io[b].x_up) + 1u;
            mp.p[b].x_seq_bytes = (long long)(rows / io[b].x_up) * Lb.cin_pad * 4;
        }
        mp.p[b].zeros = h->d_zeros;
        mp.p[b].slope_out = slope_out;
        mp.p[b].cout_real = Lb.cout_pad;
        if (f32) mp.p[b].w16 = reinterpret_cast<const bf16x8*>(Lb.d_w32);
        const double pos = (double)nseq * rows;
        flops += 2.0 * pos * Lb.cin * Lb.cout * Lb.K * zr.n;
        bytes += 4.0 * (pos * Lb.cin_pad * (io[b].x_slope >= 0.f ? io[b].x_n : 1) / std::max(1, io[b].x_up) + pos * Lb.cout_total * ((io[b].res ? 1 : 0) + (io[b].y ? 1 : 0) + (io[b].ys ? 1 : 0)) +
                        (double)Lb.cin * Lb.cout * Lb.K);
    }
    const size_t buf_bytes = round_up_sz((size_t)(TM + halo_all) * RB, 1024);
    const bool dout = f32 && tc.KS == 1;
    const size_t lds = 2 * buf_bytes + (dout ? 0 : out_buf_bytes(tc));
    mp.n_branches = nbr;
    mp.nseq_tiles = nseq * ((rows + TM - 1) / TM);
    mp.ngroups = (L0.n_blocks32 + tc.WN * tc.NB - 1) / (tc.WN * tc.NB);
    mp.total_tiles = nbr * zr.n * mp.ngroups * mp.nseq_tiles;
    mp.buf_bytes = (int)buf_bytes;
    mp.trace = nullptr;
    mp.zrep = zr.n;
    mp.zs_x = zr.zs_x;
    mp.zs_w = zr.zs_w;
    mp.zs_y = zr.zs_y;
    mp.zs_b = zr.zs_b;
    dim3 grid((unsigned)std::min(mp.total_tiles, h->num_cus), 1, 1);  // persistent: one workgroup per CU walks its tile list
    {   // one round of tiles and at least twice the activations in weight bytes (batch 8 measured neutral at 1.8 x): XCD-contiguous tile order (MultiConvParams::xcd_order)
        double wbytes = 0.0;
        for (int b = 0; b < nbr; ++b) wbytes += 4.0 * layers[b]->cin_pad * layers[b]->cout_total * layers[b]->ntaps * zr.n;
        const double abytes = 4.0 * nseq * rows * L0.cin_pad * nbr * zr.n;
        mp.xcd_order = (h->xcd_order && mp.total_tiles <= h->num_cus && mp.total_tiles >= 16 && wbytes > 2.0 * abytes) ? 1 : 0;
    }
    // every input row is staged once per channel group: more than two groups (upsampler 0's ten, a 1024-wide GEMM's eight) re-read it from the L2
    // instead of streaming it past the cache (r06a: 2.7 x / 3-7 x the algorithmic bytes fetched by those launches with non-temporal loads)
    mp.stage_cached = mp.ngroups > 2 ? 1 : 0;
    // One round of tiles of ONE layer whose weights do not fit an XCD's 4-MB L2 (the discriminators' 1024-wide GEMM-form layers: 21 MB of weights,
    // 17 row tiles x 8 channel groups): dealt out round-robin every XCD streams ALL the weights through its L2 (335 MB per launch); here every XCD
    // gets a BLOCK of (row tiles x channel groups), so that both operands are re-used inside the XCD by workgroups that run in lock step — the
    // split of the 8 XCDs into (pr x pg) that minimises row tiles / pr + channel groups / pg.  Workgroup w is dispatched to XCD w % 8
    // (MI355X_MICROARCH.md: round-robin; an assumption for speed only — any mapping is correct).
    if (mp.total_tiles <= h->num_cus && nbr == 1 && zr.n == 1 && mp.ngroups >= 2 && mp.nseq_tiles >= 8 && !h->shared_chip) {
        const double wbytes1 = 4.0 * L0.cin_pad * L0.cout_total * L0.ntaps;
        if (wbytes1 > 6.0e6) {
            const int R = mp.nseq_tiles, Gc = mp.ngroups;
            int bpr = 1, bpg = 8;
            double bcost = 1e300;
            for (int pr = 1; pr <= 8; pr *= 2) {
                const int pg = 8 / pr;
                if (pr > R || pg > Gc) continue;
                const double c = (double)((R + pr - 1) / pr) + (double)((Gc + pg - 1) / pg);
                if (c < bcost) bcost = c, bpr = pr, bpg = pg;
            }
            std::vector<std::vector<int>> per_xcd(8);
            size_t maxblock = 0;
            for (int x = 0; x < 8; ++x) {
                const int xr = x / bpg, xg = x % bpg;
                const int r0 = R * xr / bpr, r1 = R * (xr + 1) / bpr, g0 = Gc * xg / bpg, g1 = Gc * (xg + 1) / bpg;
                for (int g2 = g0; g2 < g1; ++g2)
                    for (int r2 = r0; r2 < r1; ++r2) per_xcd[(size_t)x].push_back(g2 * R + r2);  // tile id = channel group * row tiles + row tile
                maxblock = std::max(maxblock, per_xcd[(size_t)x].size());
            }
            if (maxblock * 8 <= (size_t)h->num_cus && bcost < (double)(R + 1)) {
                std::vector<std::vector<int>> lists(maxblock * 8);
                for (int x = 0; x < 8; ++x)
                    for (size_t l = 0; l < per_xcd[(size_t)x].size(); ++l) lists[l * 8 + (size_t)x].push_back(per_xcd[(size_t)x][l]);
                const std::string key = std::string(f32 ? "f" : "c") + "|x2d|" + L0.name + "|" + std::to_string(R) + "x" + std::to_string(Gc) + "t" + std::to_string(TM) +
                                        "w" + std::to_string(tc.WN) + "k" + std::to_string(tc.KS) + (tc.NB == 2 ? "b" : "");
                grid = dim3((unsigned)lists.size(), 1, 1);
                mp.xcd_order = 0;
                int rc2 = put_schedule(h, key, lists, mp.total_tiles, stream, &mp.sched_start, &mp.sched_tiles);
                if (rc2 != HIFICAR_OK) return rc2;
            }
        }
    }
    if (!mp.sched_start && h->use_lpt && mp.total_tiles > (int)grid.x) {
        std::vector<double> costs((size_t)mp.total_tiles);
        const int tpb = mp.ngroups * mp.nseq_tiles;
        std::string key = f32 ? "f" : "c";
        for (int b = 0; b < nbr; ++b) {
            key += "|" + layers[b]->name;
            for (int i = 0; i < tpb * zr.n; ++i) costs[(size_t)b * tpb * zr.n + i] = layers[b]->ntaps + 1.0;  // + fixed per-tile overhead
        }
        if (zr.n > 1) key += "z" + std::to_string(zr.n);
        key += "|" + std::to_string(nseq) + "x" + std::to_string((rows + TM - 1) / TM) + "t" + std::to_string(TM) + "w" + std::to_string(tc.WN) +
               "k" + std::to_string(tc.KS) + (tc.NB == 2 ? "b" : "");
        int rc2 = get_schedule(h, key, costs, (int)grid.x, stream, &mp.sched_start, &mp.sched_tiles);
        if (rc2 != HIFICAR_OK) return rc2;
    }
    char kname[96];
    if (dout) snprintf(kname, sizeof(kname), "conv_f32do_kernel<%d,%d,%d,%d>", tc.MI, tc.WM, tc.WN, nc16);
    else if (tc.KS == 4) snprintf(kname, sizeof(kname), "%s<%d,%d>", f32 ? "conv_sk_f32_kernel" : "conv_sk_bf16x3_kernel", tc.MI, nc16);
    else if (tc.NB == 2) snprintf(kname, sizeof(kname), "conv_bf16x3nb_kernel<%d,%d,%d,%d>", tc.MI, tc.WM, tc.WN, nc16);
    else snprintf(kname, sizeof(kname), "conv_bf16x3_kernel<%d,%d,%d,%d>", tc.MI, tc.WM, tc.WN, nc16);
    if (h->profile_detail) {  // per-layer rows in the profile (tools/layer_profile.py)
        const size_t n = strlen(kname);
        snprintf(kname + n, sizeof(kname) - n, "|%s x%d", L0.name.c_str(), nbr);
    }
    ProfScope prof(h, stream, kname, flops, bytes);
    const ConvShape shape = {dout ? kConvF32do : tc.KS == 4 ? (f32 ? kConvSkF32 : kConvSkBf16x3) : tc.NB == 2 ? kConvBf16x3nb : kConvBf16x3,
                             tc.MI, tc.WM, tc.WN, nc16};
    const hipError_t e = conv_launch(shape, &mp, grid, lds, stream);
    if (e != hipSuccess) return fail(HIFICAR_E_HIP, "conv launch (%s, %s) failed: %s", L0.name.c_str(), kname, hipGetErrorString(e));
    return HIFICAR_OK;
}

// Fused conv1 -> LeakyReLU -> conv2 (+ residual) for C = 32 / 64 (conv_pair_bf16x3_kernel / conv_pair_f32_kernel).
struct PairIOB {
    const float* xf;  // fp32 pre-activation input of conv1 (activated + split while staging), or null when xs is given
    const char* xs;   // split input of conv1
    const float* res; // fp32 residual
    float* y;         // fp32 output (may alias res)
    char* ys;         // split copy of LeakyReLU(y) for the next pair (must NOT alias xs: neighbouring tiles read xs halos), or null
};

// Launches with too few 512-row tiles to fill the chip at C = 32 in exact fp32 (batch 8 with chunks of 25 frames): 128-row tiles (MI = 1)
// — a tile's serial MFMA chain is a quarter as long and four times as many workgroups have work, so the fused pair beats two dependent
// launches there (measured batch 8: +2.7 % end to end).  Below ~8000 rows (batch 1: 2000) the split-K layer-by-layer launches, whose chains
// are shorter still, stay ahead (measured batch 1: -3.6 % with the fused form), so those keep running layer by layer.
static bool pair_small_tiles(const hificar_handle* h, int C, int k2, int nseq, int rows) {
    // (HIFICAR_KSPLIT=0, the batch-invariant mode: no launch-size-dependent forms at small sizes)
    if (!h->pair_small || h->ksplit == 0 || C != 32 || h->precision != HIFICAR_PREC_F32 || nseq <= 0 || (long long)nseq * rows < 8000) return false;
    const int tmo = 4 * 4 * 32 - (k2 - 1);
    return 3LL * nseq * ((rows + tmo - 1) / tmo) < h->num_cus;
}

// nseq / rows: the launch the pair would run in (0 / 0: only the static conditions)
static bool pair_eligible(const hificar_handle* h, const ConvLayer& a, const ConvLayer& b, int nseq = 0, int rows = 0) {
    if (!(h->use_pair && a.d_w16c && b.d_w16c && a.d_w32c && b.d_w32c && a.cin == b.cin && a.ntaps >= 2 &&
          b.ntaps >= 2 && b.dilation == 1 && a.K == b.K))
        return false;
    // the LDS budget of launch_pair (wide dilations x long kernels do not fit: those pairs run layer by layer)
    const int C = a.cin, TMc = (C == 64 ? 2 : 4) * 4 * 32;
    const size_t in_bytes = round_up_sz((size_t)(TMc + a.off_max - a.off_min) * C * 4, 1024);
    const size_t ts_bytes = std::max<size_t>((size_t)(TMc + 16) * C * 4, (size_t)TMc * (C + 4) * sizeof(float));
    if (in_bytes + ts_bytes > 160 * 1024) return false;
    if (nseq > 0) {
        // The fused kernel's tiles are tall (TMc conv1 rows for TMc - (k-1) output rows).  (i) A launch with few tiles leaves most
        // CUs idle behind long serial tiles: small batches run layer by layer.  (ii) Tile quantisation: 1000 rows at k = 11 need 5
        // tiles of 256 (28 % extra conv1 work); the exact-fp32 arithmetic gains little from fusion (its layer-by-layer kernels are
        // matrix-pipe-bound already), so it only fuses when the waste is small; the bf16x3 ones are memory-path-bound and gain more.
        if (pair_small_tiles(h, C, b.ntaps, nseq, rows)) return true;
        const int tmo = TMc - (b.ntaps - 1);
        const long long tiles = (long long)nseq * ((rows + tmo - 1) / tmo);
        if (3 * tiles < h->num_cus) return false;
        const double waste = (double)((rows + tmo - 1) / tmo) * TMc / rows;
        if (waste > (h->precision == HIFICAR_PREC_F32 ? 1.12 : 1.35)) return false;
    }
    return true;
}

static int launch_pair(hificar_handle* h, const ConvLayer* const* l1, const ConvLayer* const* l2, int nbr, int nseq, int rows,
                              const PairIOB* io, float slope, const Ragged& rg, hipStream_t stream) {
    const int C = l1[0]->cin;
    bool small = true;
    for (int b = 0; b < nbr; ++b) small = small && pair_small_tiles(h, C, l2[b]->ntaps, nseq, rows);
    const int MI = small ? 1 : 4, WM = C == 64 ? 2 : 4;
    const int TMc = WM * MI * 32, RB = C * 4;
    const bool f32 = h->precision == HIFICAR_PREC_F32;
    PairParams pp;
    memset(&pp, 0, sizeof(pp));
    int halo_max = 0;
    double flops = 0.0, bytes = 0.0;
    int tile = 0;
    for (int b = 0; b < nbr; ++b) {
        const ConvLayer& A = *l1[b];
        const ConvLayer& B = *l2[b];
        fill_params(pp.p1[b], A, rows, TMc, nullptr, nullptr, rg);
        fill_params(pp.p2[b], B, rows, TMc, io[b].res, io[b].y, rg);
        pp.p1[b].w16 = f32 ? reinterpret_cast<const bf16x8*>(A.d_w32c) : reinterpret_cast<const bf16x8*>(A.d_w16c);
        pp.p2[b].w16 = f32 ? reinterpret_cast<const bf16x8*>(B.d_w32c) : reinterpret_cast<const bf16x8*>(B.d_w16c);
        pp.p1[b].xs = io[b].xs;
        pp.p1[b].xf = io[b].xf;
        pp.p1[b].slope_in = slope;
        pp.p1[b].zeros = h->d_zeros;
        pp.p2[b].ys = io[b].ys;
        pp.p2[b].slope_out = slope;
        pp.p2[b].cout_real = C;
        halo_max = std::max(halo_max, A.off_max - A.off_min);
        const int tmo = TMc - (B.ntaps - 1);
        pp.tiles_per_seq[b] = (rows + tmo - 1) / tmo;
        pp.tile_start[b] = tile;
        tile += nseq * pp.tiles_per_seq[b];
        const double pos = (double)nseq * rows;
        flops += 2.0 * pos * C * C * (A.K + B.K);
        bytes += 4.0 * (pos * C * (3 + (io[b].ys ? 1 : 0)) + (double)C * C * (A.K + B.K));
    }
    pp.tile_start[nbr] = tile;
    pp.n_branches = nbr;
    pp.nseq = nseq;
    pp.in_bytes = (int)round_up_sz((size_t)(TMc + halo_max) * RB, 1024);
    pp.ts_bytes = (int)std::max<size_t>((size_t)(TMc + 16) * RB, (size_t)TMc * (C + 4) * sizeof(float));  // TS and out-buffer alias
    pp.slope_mid = slope;
    pp.trace = nullptr;
    const size_t lds = (size_t)pp.in_bytes + pp.ts_bytes;
    if (lds > 160 * 1024) return fail(HIFICAR_E_INVALID, "internal: pair kernel LDS too large (%zu)", lds);
    dim3 grid((unsigned)std::min(tile, h->num_cus), 1, 1);
    if (h->use_lpt && tile > (int)grid.x) {
        std::vector<double> costs((size_t)tile);
        std::string key = "p";
        for (int b = 0; b < nbr; ++b) {
            key += "|" + l1[b]->name;
            for (int i = pp.tile_start[b]; i < pp.tile_start[b + 1]; ++i) costs[i] = l1[b]->ntaps + l2[b]->ntaps + 2.0;
        }
        key += "|" + std::to_string(nseq);
        for (int b = 0; b < nbr; ++b) key += "x" + std::to_string(pp.tiles_per_seq[b]);
        int rc2 = get_schedule(h, key, costs, (int)grid.x, stream, &pp.sched_start, &pp.sched_tiles);
        if (rc2 != HIFICAR_OK) return rc2;
    }
    char kname[96];
    snprintf(kname, sizeof(kname), "%s<%d,%d,%d,%d>", f32 ? "conv_pair_f32_kernel" : "conv_pair_bf16x3_kernel", MI, WM, 4 / WM, C / 16);
    if (h->profile_detail) {
        const size_t n = strlen(kname);
        snprintf(kname + n, sizeof(kname) - n, "|%s x%d", l1[0]->name.c_str(), nbr);
    }
    ProfScope prof(h, stream, kname, flops, bytes);
    const ConvShape shape = {f32 ? kPairF32 : kPairBf16x3, MI, WM, 4 / WM, C / 16};
    const hipError_t e = conv_launch(shape, &pp, grid, lds, stream);
    if (e != hipSuccess) return fail(HIFICAR_E_HIP, "pair launch (%s) failed: %s", l1[0]->name.c_str(), hipGetErrorString(e));
    return HIFICAR_OK;
}

// ------------------------------------------------------------------------------------------------
// debug taps: copy a named intermediate (channels-last, padded pitch) into the caller's buffer in the reference's (B, C, L) layout
// ------------------------------------------------------------------------------------------------
extern "C" int hificar_debug_tap(hificar_handle* h, const char* name, float* dst, size_t capacity) {
    if (!h) return fail(HIFICAR_E_INVALID, "null handle");
    if (!name) {
        h->taps.clear();
        return HIFICAR_OK;
    }
    if (!dst) {
        h->taps.erase(name);
        return HIFICAR_OK;
    }
    h->taps[name] = {dst, capacity};
    return HIFICAR_OK;
}

static int emit_tap(hificar_handle* h, const std::string& name, const void* src, int pitch, int c0, int C, int nseq, int rows, int split,
                    hipStream_t stream, int src_rows = 0) {
    auto it = h->taps.find(name);
    if (it == h->taps.end()) return HIFICAR_OK;
    const size_t need = (size_t)nseq * C * rows;
    if (it->second.cap < need) return fail(HIFICAR_E_INVALID, "debug tap '%s' needs %zu floats, buffer has %zu", name.c_str(), need, it->second.cap);
    TapParams tp;
    tp.src = src;
    tp.dst = it->second.dst;
    tp.pitch = pitch;
    tp.c0 = c0;
    tp.C = C;
    tp.rows = rows;
    tp.src_rows = src_rows ? src_rows : rows;
    tp.total = (long long)need;
    tp.split = split;
    const unsigned blocks = (unsigned)std::min<long long>((tp.total + 255) / 256, 4096);
    hipLaunchKernelGGL(tap_copy_kernel, dim3(blocks), dim3(256), 0, stream, tp);
    HIP_TRY(hipGetLastError());
    return HIFICAR_OK;
}

static bool tap_wanted(const hificar_handle* h, const std::string& name) { return h->taps.count(name) != 0; }

// LeakyReLU(0.01) + Conv1d(C -> 1, k) + tanh on the mean of `nin` fp32 inputs (hifigan.py:146-159, 231; gblock_gen.py:71-93, 131)
static int launch_output_conv(hificar_handle* h, const float* const* fin, int nin, int Cpad, int rows, int B, int T, float* out,
                              int64_t out_bstride, const int32_t* seq_len, const Ragged& rg, const int2* slots, hipStream_t stream) {
    const hificar_config& cfg = h->cfg;
    OutConvParams op;
    memset(&op, 0, sizeof(op));
    op.x0 = fin[0];
    op.x1 = nin > 1 ? fin[1] : nullptr;
    op.x2 = nin > 2 ? fin[2] : nullptr;
    op.x3 = nin > 3 ? fin[3] : nullptr;
    op.nin = nin;
    op.w = h->d_out_w;
    op.bias = h->out_bias;
    op.bias_ptr = h->d_out_bias;
    op.out = out;
    op.out_bstride = out_bstride;
    op.L = rows;
    op.C = Cpad;
    op.K = cfg.kernel_size;
    op.slope = 0.01f;
    op.use_tanh = cfg.use_tanh;
    op.seq_len = seq_len;
    op.len_const = seq_len ? -1 : rg.const_len;
    op.len_f0 = rg.f0;
    op.len_max = T;
    op.len_mul = rows / T;
    op.slots = slots;
    op.hop = h->hop;
    // samples per workgroup: 256, or what a 64-KB LDS tile of (TR + K - 1) rows x (C + 1) floats + the weights allows
    const long long fit = ((long long)64 * 1024 / 4 - (long long)op.K * op.C) / (op.C + 1) - (op.K - 1);
    if (fit < 1) return fail(HIFICAR_E_INVALID, "output conv: %d channels x kernel %d does not fit the LDS tile", op.C, op.K);
    op.TR = (int)std::min<long long>(256, fit);
    const size_t lds = ((size_t)(op.TR + op.K - 1) * (op.C + 1) + (size_t)op.K * op.C) * sizeof(float);
    {
        const double pos = (double)B * rows;
        ProfScope prof(h, stream, "output_conv_kernel", 2.0 * pos * op.C * op.K, 4.0 * pos * (op.C * nin + 1));
        hipLaunchKernelGGL(output_conv_kernel, dim3((rows + op.TR - 1) / op.TR, B), dim3(256), lds, stream, op);
    }
    HIP_TRY(hipGetLastError());
    return HIFICAR_OK;
}

struct Tape;
struct Cond;
struct Workspace;
// GBlockGenerator body of a forward: input conv -> GBlocks -> output conv (hificar_gblock.hip.inc)
static int gblock_forward(hificar_handle* h, float* out, int64_t out_bstride, int B, int T, const Workspace& ws, hipStream_t stream,
                          const int32_t* seq_len, const Ragged& rg, const int2* slots, const Tape* tp);

// One generator forward on B sequences of T frames.
//   c: element (b, ch, t) at c[b*c_bstride + ch*c_cstride + t];  prev: (b, i) at prev[b*prev_bstride + i] or null
//   out: sample (b, n) at out[b*out_bstride + n]
static int forward_impl(hificar_handle* h, const float* c, int64_t c_bstride, int64_t c_cstride, const float* prev,
                        int64_t prev_bstride, float* out, int64_t out_bstride, int B, int T, const Workspace& ws,
                        hipStream_t stream, const int32_t* seq_len = nullptr, int f0 = 0, const int2* slots = nullptr, int T_valid = -1,
                        const Cond& cond = Cond(), const Tape* tp = nullptr) {
    // T: frames the launches cover; T_valid (<= T, default T): frames that exist in c / out (bucketed non-AR lengths)
    const hificar_config& cfg = h->cfg;
    if (T_valid < 0) T_valid = T;
    Ragged rg;
    rg.seq_len = seq_len;
    rg.const_len = T_valid < T ? f0 + T_valid : -1;
    rg.f0 = f0;
    rg.frames = T;
    // 1. front end
    FrontParams fp;
    memset(&fp, 0, sizeof(fp));
    fp.c = c;
    fp.c_bstride = c_bstride;
    fp.c_cstride = c_cstride;
    fp.prev = prev;
    fp.prev_bstride = prev_bstride;
    fp.slots = slots;
    fp.valid = seq_len;
    fp.hop = h->hop;
    const bool f32 = h->precision == HIFICAR_PREC_F32;
    fp.xin = f32 ? (tp ? tp->xin : ws.xin) : nullptr;
    fp.xin_s = f32 ? nullptr : reinterpret_cast<char*>(ws.xin);
    fp.mlp_tape = tp ? tp->mlp : nullptr;
    fp.T = T;
    fp.t_valid = T_valid;
    if (cfg.use_spk_id) {
        fp.spk_id = cond.spk_id;
        fp.spk_emb = h->d_spk_emb;
        fp.spk_w = h->d_spk_w;
        fp.spk_b = h->d_spk_b;
        fp.spk_e = cfg.spk_emb_size;
    }
    if (cfg.use_ph) {
        fp.ph = cond.ph;
        fp.ph_stride = cond.ph_stride;
        fp.ph_emb = h->d_ph_emb;
        fp.ph_e = cfg.ph_emb_size;
    }
    fp.cf = h->cf;
    fp.cin_pad = h->cin_pad;
    fp.use_ar = cfg.use_ar;
    fp.ar_input = cfg.ar_input;
    fp.ar_hidden = cfg.ar_hidden;
    fp.ar_output = cfg.ar_output;
    for (int l = 0; l < 5; ++l) {
        fp.wt[l] = h->d_mlp_w[l];
        fp.bs[l] = h->d_mlp_b[l];
    }
    {
        const double mlp_macs = cfg.use_ar ? (double)cfg.ar_input * cfg.ar_hidden + 3.0 * cfg.ar_hidden * cfg.ar_hidden +
                                                 (double)cfg.ar_hidden * cfg.ar_output : 0.0;
        ProfScope prof(h, stream, "front_kernel", 2.0 * B * mlp_macs, 4.0 * B * (mlp_macs + (double)T * (h->cf + h->cin_pad)));
        hipLaunchKernelGGL(front_kernel, dim3(B), dim3(kFrontThreads), 0, stream, fp);
    }
    HIP_TRY(hipGetLastError());
    if (h->arch == 1) return gblock_forward(h, out, out_bstride, B, T, ws, stream, seq_len, rg, slots, tp);

    int rc;
    int rows = T;
    const bool tapping = !h->taps.empty();
    bool tap_convs1 = false;  // a conv1 output is wanted: those pairs run layer by layer (the fused kernel keeps it in LDS)
    if (tapping) {
        for (auto& kv : h->taps) tap_convs1 = tap_convs1 || kv.first.find(".convs1.") != std::string::npos;
        // pre-activation copies that the normal path never writes: 3 stage-sized scratch buffers
        const size_t need = kMaxBlk * stage_elems(h, B, T) + (size_t)B * T * stage_pad(cfg, 0);
        if (need > h->tap_scratch_elems) {
            if (h->tap_scratch) HIP_TRY(hipFree(h->tap_scratch));
            h->tap_scratch = nullptr;
            h->tap_scratch_elems = 0;
            void* p = nullptr;
            HIP_TRY(hipMalloc(&p, need * sizeof(float)));
            h->tap_scratch = static_cast<float*>(p);
            h->tap_scratch_elems = need;
        }
        if (cfg.use_ar && (rc = emit_tap(h, "ar_feats", ws.xin, f32 ? h->cin_pad : -h->cin_pad, h->cf, cfg.ar_output, B, 1, f32 ? 0 : 1, stream, T)) != HIFICAR_OK)
            return rc;
    }
    const size_t tap_se = tapping ? stage_elems(h, B, T) : 0;
    const int nbk = cfg.n_blocks;
    // residual blocks of a stage run side by side, heaviest kernel size first
    int order[kMaxBlk];
    for (int j = 0; j < kMaxBlk; ++j) order[j] = j;
    std::sort(order, order + nbk, [&](int a, int b) { return cfg.resblock_kernel_sizes[a] > cfg.resblock_kernel_sizes[b]; });
    int max_d = 0;
    for (int j = 0; j < nbk; ++j) max_d = std::max(max_d, cfg.n_dilations[j]);
    int rows_of_launch = 0;  // (conv_n: the stage's row count at the time of the call)
    const bool add_convs = cfg.use_additional_convs != 0;  // false: a ResBlock layer is x = x + conv1(LeakyReLU(x)) (residual_block.py:217-221)
    // a launch carries up to three branches (MultiConvParams / PairParams): a fourth residual block rides in a second launch
    auto conv_n = [&](const ConvLayer* const* lay, int n, const ConvIO* io) -> int {
        for (int q0 = 0; q0 < n; q0 += 3) {
            const int r = launch_conv(h, lay + q0, std::min(3, n - q0), B, rows_of_launch, io + q0, cfg.lrelu_slope, rg, stream);
            if (r != HIFICAR_OK) return r;
        }
        return HIFICAR_OK;
    };

    const float* fin[kMaxBlk] = {ws.x[0], ws.x[1], ws.x[2], ws.x[3]};  // where each branch's ResBlock output of the current stage lives
    {
        // Activations travel between layers already activated — split rows (bf16x3) or plain fp32 rows (exact fp32), the
        // "_s" buffers — and are staged by LDS-DMA; the layer's own fp32 value only where a residual / the MRF mean needs it
        char* xin_s = tp ? reinterpret_cast<char*>(tp->xin) : reinterpret_cast<char*>(ws.xin);
        char* h0_s = tp ? tp->h0_s : reinterpret_cast<char*>(ws.h0);
        char* xt_s[kMaxBlk] = {reinterpret_cast<char*>(ws.xt[0]), reinterpret_cast<char*>(ws.xt[1]), reinterpret_cast<char*>(ws.xt[2]),
                               reinterpret_cast<char*>(ws.xt[3])};
        {   // 2. input conv (no activation in front of it: hifigan.py:221); its consumer applies LeakyReLU(slope)
            const ConvLayer* lay[1] = {&h->input_conv};
            float* y_tap = tap_wanted(h, "input_conv") ? h->tap_scratch + kMaxBlk * tap_se : nullptr;
            const ConvIO io[1] = {{xin_s, nullptr, y_tap, h0_s}};
            if ((rc = launch_conv(h, lay, 1, B, T, io, cfg.lrelu_slope, rg, stream)) != HIFICAR_OK) return rc;
            if (y_tap && (rc = emit_tap(h, "input_conv", y_tap, stage_pad(cfg, 0), 0, cfg.channels, B, T, 0, stream)) != HIFICAR_OK) return rc;
        }
        for (int i = 0; i < cfg.n_stages; ++i) {
            const char* up_in = h0_s;
            // MRF mean of the previous stage (hifigan.py:226-230).  Exact fp32 inference: folded into the upsampler's staging (ConvIO::x_more — its
            // loader waves read the blocks' fp32 streams, sum, divide, activate), so no launch and no buffer for the mean exist.  Training (the tape keeps
            // the activated mean for the upsampler's weight gradient) and bf16x3 (split rows): mrf_split_kernel.
            const bool fold_mrf = i > 0 && f32 && !tp;
            if (i > 0 && !fold_mrf) {
                MrfSplitParams mq;
                memset(&mq, 0, sizeof(mq));
                mq.x0 = fin[0];
                mq.x1 = nbk > 1 ? fin[1] : nullptr;
                mq.x2 = nbk > 2 ? fin[2] : nullptr;
                mq.x3 = nbk > 3 ? fin[3] : nullptr;
                mq.out = fin[0] == ws.xt[0] ? ws.x_s[0] : xt_s[0];  // a buffer none of the inputs lives in
                if (tp) mq.out = tp->upin_s[i];
                mq.nin = nbk;
                mq.C = stage_pad(cfg, i);
                mq.rows = (long long)B * rows;
                mq.slope = cfg.lrelu_slope;
                mq.f32 = f32 ? 1 : 0;
                const long long units = mq.rows * (mq.C / 8);
                const unsigned blocks = (unsigned)std::min<long long>((units + 255) / 256, 8LL * h->num_cus);
                {
                    ProfScope prof(h, stream, "mrf_split_kernel", 0.0, 4.0 * mq.rows * mq.C * (nbk + 1));
                    hipLaunchKernelGGL(mrf_split_kernel, dim3(blocks), dim3(256), 0, stream, mq);
                }
                HIP_TRY(hipGetLastError());
                up_in = mq.out;
            }
            // Narrow stage whose every layer pair runs in the fused kernel: the residual stream stays fp32-only (the pair
            // kernel activates + splits its input while staging), ping-ponging between x[j] and xt[j]; no activated copies
            // are written at all.  Otherwise: activated copies ("_s") travel next to the fp32 stream.
            bool all_pairs = !tap_convs1 && !tp && add_convs;
            for (int j = 0; j < nbk && all_pairs; ++j)
                for (int d = 0; d < cfg.n_dilations[j]; ++d) {
                    const int ci = conv_index(h, i, j, d);
                    all_pairs = all_pairs && pair_eligible(h, h->convs1[ci], h->convs2[ci], B, rows * cfg.upsample_scales[i]);
                }
            {   // LeakyReLU + ConvTranspose1d (hifigan.py:224): fp32 u (first residual) (+ activated copy: first conv input)
                const ConvLayer* lay[1] = {&h->ups[i]};
                ConvIO io[1] = {{up_in, nullptr, ws.u, all_pairs ? nullptr : (tp ? tp->u_s[i] : ws.u_s)}};
                if (fold_mrf) {
                    io[0].xs = reinterpret_cast<const char*>(fin[0]);
                    io[0].x_slope = cfg.lrelu_slope;
                    io[0].x_n = nbk;
                    for (int j = 1; j < nbk; ++j) io[0].x_more[j - 1] = fin[j];
                }
                if ((rc = launch_conv(h, lay, 1, B, rows, io, cfg.lrelu_slope, rg, stream)) != HIFICAR_OK) return rc;
            }
            rows *= cfg.upsample_scales[i];
            const int Cs = stage_channels(cfg, i + 1), Cp = stage_pad(cfg, i + 1);
            if (tapping && (rc = emit_tap(h, "upsamples." + std::to_string(i), ws.u, Cp, 0, Cs, B, rows, 0, stream)) != HIFICAR_OK) return rc;
            auto tap_block = [&](int j, int d, const float* xcur) -> int {  // residual stream of block j after dilation d
                if (!tapping) return HIFICAR_OK;
                const std::string base = "blocks." + std::to_string(i * nbk + j);
                int r = emit_tap(h, base + ".x." + std::to_string(d), xcur, Cp, 0, Cs, B, rows, 0, stream);
                if (r == HIFICAR_OK && d + 1 == cfg.n_dilations[j]) r = emit_tap(h, base, xcur, Cp, 0, Cs, B, rows, 0, stream);
                return r;
            };
            if (all_pairs) {
                const float* cur_f[kMaxBlk] = {ws.u, ws.u, ws.u, ws.u};
                for (int d = 0; d < max_d; ++d) {  // residual_block.py:217-221
                    const ConvLayer* l1[kMaxBlk];
                    const ConvLayer* l2[kMaxBlk];
                    PairIOB iop[kMaxBlk];
                    int n = 0;
                    for (int oj = 0; oj < nbk; ++oj) {
                        const int j = order[oj];
                        if (d >= cfg.n_dilations[j]) continue;
                        const int ci = conv_index(h, i, j, d);
                        l1[n] = &h->convs1[ci];
                        l2[n] = &h->convs2[ci];
                        // a tile's output pass must not overwrite rows a neighbouring tile still reads as halo: out != in
                        float* out_f = cur_f[j] == ws.x[j] ? ws.xt[j] : ws.x[j];
                        iop[n] = {cur_f[j], nullptr, cur_f[j], out_f, nullptr};
                        cur_f[j] = out_f;
                        ++n;
                    }
                    for (int q0 = 0; q0 < n; q0 += 3)
                        if ((rc = launch_pair(h, l1 + q0, l2 + q0, std::min(3, n - q0), B, rows, iop + q0, cfg.lrelu_slope, rg, stream)) != HIFICAR_OK) return rc;
                    for (int j = 0; j < nbk; ++j)
                        if (d < cfg.n_dilations[j] && (rc = tap_block(j, d, cur_f[j])) != HIFICAR_OK) return rc;
                }
                for (int j = 0; j < nbk; ++j) fin[j] = cur_f[j];
                continue;
            }
            // fp32 residual streams; training keeps the LAST stage's (the output conv's backward needs the MRF mean) in the tape
            float* xres[kMaxBlk] = {ws.x[0], ws.x[1], ws.x[2], ws.x[3]};
            if (tp && i + 1 == cfg.n_stages)
                for (int j = 0; j < nbk; ++j) xres[j] = tp->fin[j];
            for (int j = 0; j < nbk; ++j) fin[j] = xres[j];
            // activated stream of each branch: where the next conv1 reads its input.  It alternates between x_s[j] and
            // xt_s[j]: a launch never writes the buffer it (or a neighbouring tile, through the halo) reads.
            const char* u_act = tp ? tp->u_s[i] : ws.u_s;
            const char* cur_s[kMaxBlk] = {u_act, u_act, u_act, u_act};
            for (int d = 0; d < max_d; ++d) {  // residual_block.py:217-221
                const ConvLayer* l1[kMaxBlk];
                const ConvLayer* l2[kMaxBlk];
                ConvIO io1[kMaxBlk], io2[kMaxBlk];
                PairIOB iop[kMaxBlk];
                int jn[kMaxBlk];
                char* pair_out[kMaxBlk];
                char* lbl_out[kMaxBlk];
                int n = 0;
                bool fuse = add_convs;
                for (int oj = 0; oj < nbk; ++oj) {
                    const int j = order[oj];
                    if (d >= cfg.n_dilations[j]) continue;
                    const int ci = conv_index(h, i, j, d);
                    l1[n] = &h->convs1[ci];
                    l2[n] = add_convs ? &h->convs2[ci] : nullptr;
                    fuse = fuse && !tap_convs1 && !tp && pair_eligible(h, *l1[n], *l2[n], B, rows);
                    const bool last = d + 1 == cfg.n_dilations[j];
                    // fused pair: cur -> the other buffer.  Layer by layer: cur -> mid -> the buffer that is not mid.
                    pair_out[n] = cur_s[j] == ws.x_s[j] ? xt_s[j] : ws.x_s[j];
                    char* mid = cur_s[j] == xt_s[j] ? ws.x_s[j] : xt_s[j];
                    lbl_out[n] = mid == xt_s[j] ? ws.x_s[j] : xt_s[j];
                    if (tp) {  // training: unique buffers, kept for the backward pass
                        mid = tp->xt_s[i][j][d];
                        lbl_out[n] = tp->x_s[i][j][d];
                    }
                    io1[n] = {cur_s[j], nullptr, tap_convs1 ? h->tap_scratch + (size_t)n * tap_se : nullptr, mid};
                    io2[n] = {mid, d == 0 ? ws.u : xres[j], xres[j], last ? nullptr : lbl_out[n]};
                    if (!add_convs) {  // one conv per layer: conv1 carries the residual epilogue; its activated output is the next layer's input
                        char* nxt = tp ? tp->x_s[i][j][d] : pair_out[n];
                        io1[n] = {cur_s[j], d == 0 ? ws.u : xres[j], xres[j], last ? nullptr : nxt};
                        lbl_out[n] = nxt;
                    }
                    iop[n] = {nullptr, cur_s[j], d == 0 ? ws.u : xres[j], xres[j], last ? nullptr : pair_out[n]};
                    jn[n] = j;
                    ++n;
                }
                if (fuse) {
                    for (int q0 = 0; q0 < n; q0 += 3)
                        if ((rc = launch_pair(h, l1 + q0, l2 + q0, std::min(3, n - q0), B, rows, iop + q0, cfg.lrelu_slope, rg, stream)) != HIFICAR_OK) return rc;
                } else if (!add_convs) {
                    rows_of_launch = rows;
                    if ((rc = conv_n(l1, n, io1)) != HIFICAR_OK) return rc;
                } else {
                    rows_of_launch = rows;
                    if ((rc = conv_n(l1, n, io1)) != HIFICAR_OK) return rc;
                    if (tap_convs1)
                        for (int q = 0; q < n; ++q)
                            if ((rc = emit_tap(h, "blocks." + std::to_string(i * nbk + jn[q]) + ".convs1." + std::to_string(d), io1[q].y, Cp, 0, Cs,
                                               B, rows, 0, stream)) != HIFICAR_OK)
                                return rc;
                    if ((rc = conv_n(l2, n, io2)) != HIFICAR_OK) return rc;
                }
                for (int q = 0; q < n; ++q)
                    if ((rc = tap_block(jn[q], d, xres[jn[q]])) != HIFICAR_OK) return rc;
                for (int q = 0; q < n; ++q) cur_s[jn[q]] = fuse ? pair_out[q] : lbl_out[q];
            }
        }
    }
    if (cfg.use_ph_loss && cond.ph_out) {  // phoneme-loss head on the last stage's MRF mean (hifigan.py:232-237)
        PhHeadParams pq;
        memset(&pq, 0, sizeof(pq));
        pq.x0 = fin[0];
        pq.x1 = nbk > 1 ? fin[1] : nullptr;
        pq.x2 = nbk > 2 ? fin[2] : nullptr;
        pq.x3 = nbk > 3 ? fin[3] : nullptr;
        pq.nin = nbk;
        pq.w = h->d_phfc_w;
        pq.bias = h->d_phfc_b;
        pq.out = cond.ph_out;
        pq.C = stage_channels(cfg, cfg.n_stages);
        pq.Cp = stage_pad(cfg, cfg.n_stages);
        pq.L = rows;
        pq.T = cond.ph_out_T;
        pq.hop = h->hop;
        pq.num_ph = cfg.num_ph;
        pq.seq_len = seq_len;
        pq.len_const = seq_len ? -1 : (T_valid < T ? T_valid : -1);
        ProfScope prof(h, stream, "ph_head_kernel", 2.0 * B * T_valid * (double)pq.C * cfg.num_ph, 4.0 * B * (double)rows * pq.Cp * nbk);
        hipLaunchKernelGGL(ph_head_kernel, dim3(T_valid, B), dim3(256), 0, stream, pq);
        HIP_TRY(hipGetLastError());
    }
    // 4. output conv: LeakyReLU(0.01) + Conv1d + tanh (hifigan.py:146-159)
    return launch_output_conv(h, fin, nbk, stage_pad(cfg, cfg.n_stages), rows, B, T, out, out_bstride, seq_len, rg, slots, stream);
}

static int check_ready(hificar_handle* h, int B, int T, void* ws, size_t ws_bytes) {
    if (!h) return fail(HIFICAR_E_INVALID, "null handle");
    if (!h->finalized) return fail(HIFICAR_E_STATE, "hificar_finalize has not been called");
    if (B < 1 || T < 1) return fail(HIFICAR_E_INVALID, "B=%d, T=%d must be positive", B, T);
    const size_t need = hificar_workspace_bytes(h, B, T);
    if (!ws || ws_bytes < need) return fail(HIFICAR_E_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, ws_bytes);
    if ((reinterpret_cast<uintptr_t>(ws) & 255) != 0) return fail(HIFICAR_E_INVALID, "workspace must be 256-byte aligned");
    return HIFICAR_OK;
}

extern "C" int hificar_forward_cond(hificar_handle* h, const float* c, const float* ar, const int32_t* spk_id, const int32_t* ph,
                                    const int32_t* lengths, float* out, float* ph_out, int B, int T, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    int rc = check_ready(h, B, T, workspace, workspace_bytes);
    if (rc != HIFICAR_OK) return rc;
    if (!c || !out) return fail(HIFICAR_E_INVALID, "hificar_forward: null tensor");
    if (h->cfg.use_ar && !ar) return fail(HIFICAR_E_INVALID, "use_ar model needs the ar context (got NULL)");
    if (h->cfg.use_spk_id && !spk_id) return fail(HIFICAR_E_INVALID, "use_spk_id model needs spk_id (got NULL)");
    if (h->cfg.use_ph && !ph) return fail(HIFICAR_E_INVALID, "use_ph model needs ph (got NULL)");
    Cond cond;
    cond.spk_id = spk_id;
    cond.ph = ph;
    cond.ph_stride = T;
    cond.ph_out = ph_out;
    cond.ph_out_T = T;
    if ((rc = enter_stream(h, static_cast<hipStream_t>(stream))) != HIFICAR_OK) return rc;
    const int Tb = bucket_frames(T);
    const Workspace ws = plan_workspace(h, B, Tb, workspace);
    return forward_impl(h, c, (int64_t)h->cf * T, T, h->cfg.use_ar ? ar : nullptr, h->cfg.ar_input, out,
                        (int64_t)h->hop * T, B, Tb, ws, static_cast<hipStream_t>(stream), lengths, 0, nullptr, T, cond);
}

extern "C" int hificar_forward_ragged(hificar_handle* h, const float* c, const float* ar, const int32_t* lengths, float* out, int B,
                                      int T, void* workspace, size_t workspace_bytes, void* stream) {
    if (h && (h->cfg.use_spk_id || h->cfg.use_ph))
        return fail(HIFICAR_E_INVALID, "this model takes spk_id / ph: call hificar_forward_cond");
    return hificar_forward_cond(h, c, ar, nullptr, nullptr, lengths, out, nullptr, B, T, workspace, workspace_bytes, stream);
}

extern "C" int hificar_forward(hificar_handle* h, const float* c, const float* ar, float* out, int B, int T, void* workspace,
                               size_t workspace_bytes, void* stream) {
    return hificar_forward_ragged(h, c, ar, nullptr, out, B, T, workspace, workspace_bytes, stream);
}

extern "C" int hificar_ar_loop_ragged(hificar_handle* h, const float* c, const int32_t* lengths, const int32_t* lengths_host,
                                      float* out, int B, int T_total, int chunk_frames, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    if (h && !h->cfg.use_ar) return fail(HIFICAR_E_INVALID, "hificar_ar_loop on a model built with use_ar=false");
    if (h && (h->cfg.use_spk_id || h->cfg.use_ph))  // the reference's ar_loop calls model(c, ar=prev) only (decode.py:72)
        return fail(HIFICAR_E_INVALID, "hificar_ar_loop: speaker / phoneme conditioned models are driven through hificar_forward_cond");
    if (chunk_frames < 1) return fail(HIFICAR_E_INVALID, "chunk_frames=%d must be positive", chunk_frames);
    int rc = check_ready(h, B, std::min(chunk_frames, std::max(T_total, 1)), workspace, workspace_bytes);
    if (rc != HIFICAR_OK) return rc;
    if (T_total < 1) return fail(HIFICAR_E_INVALID, "T_total=%d must be positive", T_total);
    if (!c || !out) return fail(HIFICAR_E_INVALID, "hificar_ar_loop: null tensor");
    if (h->cfg.ar_input > h->hop * chunk_frames && T_total > chunk_frames)
        return fail(HIFICAR_E_INVALID, "ar_input (%d) > chunk audio length (%d): the reference loop (decode.py:79-81) is ill-formed there",
                    h->cfg.ar_input, h->hop * chunk_frames);
    if ((rc = enter_stream(h, static_cast<hipStream_t>(stream))) != HIFICAR_OK) return rc;
    const int64_t out_bstride = (int64_t)h->hop * T_total;
    if (lengths_host && !lengths) return fail(HIFICAR_E_INVALID, "hificar_ar_loop_ragged: lengths_host without the device copy");
    if (lengths_host)
        for (int b = 0; b < B; ++b)
            if (lengths_host[b] < 0 || lengths_host[b] > T_total)
                return fail(HIFICAR_E_INVALID, "lengths[%d]=%d outside [0, %d]", b, lengths_host[b], T_total);
    // Mid-size batches on TWO streams (round 4).  Utterances do not depend on each other, and a step is a chain of 34 dependent launches whose tile
    // lists quantise badly between batch 17 and 62 (a 25-frame chunk is ONE 128-row tile per utterance at the widest stage: 6 B tiles of weight
    // 11 : 7 : 3 for 256 workgroups — batch 44 to 64 all take the same 11 units): the two halves of the batch run their own chains on two streams
    // and each half's idle workgroups, launch latencies and tails are filled by the other half's kernels.  Measured (10-s clips, chunk 25, fp32,
    // single / dual ms per step): batch 18 108.2 / 103.4, 22 130.3 / 122.6, 28 146.3 / 135.3, 32 153.1 / 142.2, 36 184.6 / 171.3, 44 224.8 / 201.5,
    // 48 225.4 / 214.7, 56 263.7 / 239.2, 60 266.2 / 258.8; below the window the halves' launches are less efficient than the whole batch's
    // (batch 8: 58.1 / 68.1, 4: 44.2 / 48.4, 16: 90.7 / 91.6), at 64 the tile lists are full (269.3 / 276.5).  Same kernels, same per-utterance
    // arithmetic up to the launch-shape dependence HIFICAR_KSPLIT=0 removes.  Not while profiling or tapping (one stream's events / scratch), not for
    // ragged batches (their steps shrink the batch prefix).
    const hipStream_t s0 = static_cast<hipStream_t>(stream);
    const int Tn_max = std::min(chunk_frames, T_total);
    const int B0 = (B + 1) / 2, B1 = B - B0;
    const size_t ws0_bytes = plan_workspace(h, B0, Tn_max, nullptr).bytes;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s0, &cap);  // (a loop being captured into a hipGraph stays on the capturing stream)
    if (!lengths && B >= 2 && B >= h->ar_dual_min && B <= h->ar_dual_max && !h->profiling && h->taps.empty() && cap == hipStreamCaptureStatusNone &&
        ws0_bytes + plan_workspace(h, B1, Tn_max, nullptr).bytes <= workspace_bytes) {
        if (!h->ar_side) {
            HIP_TRY(hipStreamCreateWithFlags(&h->ar_side, hipStreamNonBlocking));
            for (hipEvent_t& e : h->ar_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        const hipStream_t s1 = h->ar_side;
        HIP_TRY(hipEventRecord(h->ar_ev[0], s0));  // fork: the side stream starts behind the caller's work so far
        HIP_TRY(hipStreamWaitEvent(s1, h->ar_ev[0], 0));
        // The halves' launches share the chip, so a launch's own exact makespan is the wrong yardstick for its tile shape (measured: with the
        // simulated makespan the halves pick shapes that fill the chip alone and batch 32 / 44 lose 5 / 4 %; choosing by workgroup-time as the
        // discriminators' engine does loses 20-30 %): they keep the closed-form estimate.  (Reset below on every path: nothing in between returns.)
        h->shared_chip = true;
        unsigned long long seen = h->sched_up_seq;
        // a tile schedule first needed by one half is uploaded on that half's stream: the other stream must not use it before it has landed
        auto publish = [&](hipStream_t from, hipStream_t to, hipEvent_t ev) -> int {
            if (h->sched_up_seq != seen) {
                HIP_TRY(hipEventRecord(ev, from));
                HIP_TRY(hipStreamWaitEvent(to, ev, 0));
                seen = h->sched_up_seq;
            }
            return HIFICAR_OK;
        };
        const float* const c1 = c + (size_t)B0 * h->cf * T_total;
        float* const out1 = out + (size_t)B0 * out_bstride;
        char* const wsp1 = static_cast<char*>(workspace) + ws0_bytes;
        for (int f0 = 0; f0 < T_total && rc == HIFICAR_OK; f0 += chunk_frames) {
            const int Tn = std::min(chunk_frames, T_total - f0);
            const int64_t pos = (int64_t)h->hop * f0;
            const int64_t back = f0 == 0 ? 0 : pos - h->cfg.ar_input;
            rc = forward_impl(h, c + f0, (int64_t)h->cf * T_total, T_total, f0 == 0 ? nullptr : out + back, out_bstride, out + pos, out_bstride, B0, Tn,
                              plan_workspace(h, B0, Tn, workspace), s0, nullptr, f0);
            if (rc == HIFICAR_OK) rc = publish(s0, s1, h->ar_ev[0]);
            if (rc == HIFICAR_OK)
                rc = forward_impl(h, c1 + f0, (int64_t)h->cf * T_total, T_total, f0 == 0 ? nullptr : out1 + back, out_bstride, out1 + pos, out_bstride, B1, Tn,
                                  plan_workspace(h, B1, Tn, wsp1), s1, nullptr, f0);
            if (rc == HIFICAR_OK) rc = publish(s1, s0, h->ar_ev[1]);
        }
        h->shared_chip = false;
        // join (also on an error return: the caller's stream must stay ordered behind what the side stream was given)
        if (hipEventRecord(h->ar_ev[1], s1) != hipSuccess || hipStreamWaitEvent(s0, h->ar_ev[1], 0) != hipSuccess)
            return rc != HIFICAR_OK ? rc : fail(HIFICAR_E_HIP, "hificar_ar_loop: joining the side stream failed");
        return rc;
    }
    for (int f0 = 0; f0 < T_total; f0 += chunk_frames) {
        const int Tn = std::min(chunk_frames, T_total - f0);
        // With the host copy of the lengths the step only covers the utterances still running: the batch prefix up to the
        // last one longer than f0 (all of them when the batch is sorted longest first).
        int Bn = B;
        if (lengths_host) {
            Bn = 0;
            for (int b = 0; b < B; ++b)
                if (lengths_host[b] > f0) Bn = b + 1;
            if (Bn == 0) break;
        }
        const int64_t pos = (int64_t)h->hop * f0;
        // prev = last ar_input samples already written for this utterance (zeros for the first chunk)
        const float* prev = f0 == 0 ? nullptr : out + pos - h->cfg.ar_input;
        rc = forward_impl(h, c + f0, (int64_t)h->cf * T_total, T_total, prev, out_bstride, out + pos, out_bstride, Bn, Tn,
                          plan_workspace(h, Bn, Tn, workspace), static_cast<hipStream_t>(stream), lengths, f0);
        if (rc != HIFICAR_OK) return rc;
    }
    return HIFICAR_OK;
}

// Packed (continuously batched) AR synthesis: N utterances, at most `batch` of them in flight; as soon as one finishes the
// next one takes its place, so every step but the last few runs a full batch whatever the lengths are.  The AR state of
// an utterance is its own waveform so far, so a "slot" exists only in the host-side step table uploaded here.
extern "C" int hificar_ar_loop_packed(hificar_handle* h, const float* c, const int32_t* lengths_host, float* out, int N, int T_max,
                                      int chunk_frames, int batch, void* workspace, size_t workspace_bytes, void* stream_) {
    if (h && !h->cfg.use_ar) return fail(HIFICAR_E_INVALID, "hificar_ar_loop on a model built with use_ar=false");
    if (h && (h->cfg.use_spk_id || h->cfg.use_ph))  // the reference's ar_loop calls model(c, ar=prev) only (decode.py:72)
        return fail(HIFICAR_E_INVALID, "hificar_ar_loop: speaker / phoneme conditioned models are driven through hificar_forward_cond");
    if (chunk_frames < 1 || batch < 1 || N < 1 || T_max < 1)
        return fail(HIFICAR_E_INVALID, "hificar_ar_loop_packed: N=%d, T_max=%d, chunk_frames=%d, batch=%d must be positive", N, T_max,
                    chunk_frames, batch);
    batch = std::min(batch, N);
    int rc = check_ready(h, batch, std::min(chunk_frames, T_max), workspace, workspace_bytes);
    if (rc != HIFICAR_OK) return rc;
    if (!c || !out || !lengths_host) return fail(HIFICAR_E_INVALID, "hificar_ar_loop_packed: null argument");
    if (h->cfg.ar_input > h->hop * chunk_frames && T_max > chunk_frames)
        return fail(HIFICAR_E_INVALID, "ar_input (%d) > chunk audio length (%d): the reference loop (decode.py:79-81) is ill-formed there",
                    h->cfg.ar_input, h->hop * chunk_frames);
    for (int u = 0; u < N; ++u)
        if (lengths_host[u] < 0 || lengths_host[u] > T_max)
            return fail(HIFICAR_E_INVALID, "lengths[%d]=%d outside [0, %d]", u, lengths_host[u], T_max);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if ((rc = enter_stream(h, stream)) != HIFICAR_OK) return rc;
    // step table: row s lists the utterances advanced by step s as (utterance, first frame) + their valid frames
    struct Step {
        int n, frames;
    };
    std::vector<Step> steps;
    std::vector<int2> slots;
    std::vector<int> valid;
    {
        std::vector<int> run, f0((size_t)N, 0);
        int next = 0;
        for (;;) {
            while ((int)run.size() < batch && next < N) {
                if (lengths_host[next] > 0) run.push_back(next);
                ++next;
            }
            if (run.empty()) break;
            // every step is launched over a full chunk (shorter last chunks are masked per sequence and their empty tiles
            // skipped): launch shapes then differ by the number of running utterances only, and their schedules stay cached
            Step st{(int)run.size(), std::min(chunk_frames, T_max)};
            for (int u : run) {
                slots.push_back(int2{u, f0[u]});
                valid.push_back(std::min(chunk_frames, lengths_host[u] - f0[u]));
            }
            slots.resize((slots.size() + 1) & ~(size_t)1);  // rows start on an even index in both arrays (int2 rows 16-byte aligned)
            valid.resize(slots.size());
            steps.push_back(st);
            std::vector<int> keep;
            for (int u : run) {
                f0[u] += chunk_frames;
                if (f0[u] < lengths_host[u]) keep.push_back(u);
            }
            run.swap(keep);
        }
    }
    if (steps.empty()) return HIFICAR_OK;
    const size_t tab_bytes = slots.size() * (sizeof(int2) + sizeof(int));
    if (!h->tab_copied) HIP_TRY(hipEventCreateWithFlags(&h->tab_copied, hipEventDisableTiming));
    else HIP_TRY(hipEventSynchronize(h->tab_copied));  // the previous upload has left the staging copy (long ago, normally)
    if (tab_bytes > h->tab_bytes) {
        if (h->d_tab) HIP_TRY(hipFree(h->d_tab));  // synchronises: no earlier call can still be reading it
        if (h->h_tab) HIP_TRY(hipHostFree(h->h_tab));
        h->d_tab = h->h_tab = nullptr;
        h->tab_bytes = 0;
        HIP_TRY(hipMalloc(&h->d_tab, tab_bytes * 2));
        HIP_TRY(hipHostMalloc(&h->h_tab, tab_bytes * 2, hipHostMallocDefault));
        h->tab_bytes = tab_bytes * 2;
    }
    int2* d_slots = static_cast<int2*>(h->d_tab);
    int* d_valid = reinterpret_cast<int*>(d_slots + slots.size());
    memcpy(h->h_tab, slots.data(), slots.size() * sizeof(int2));
    memcpy(static_cast<char*>(h->h_tab) + slots.size() * sizeof(int2), valid.data(), valid.size() * sizeof(int));
    // one asynchronous upload, ordered on the stream behind any earlier call that still reads the table
    HIP_TRY(hipMemcpyAsync(h->d_tab, h->h_tab, tab_bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(h->tab_copied, stream));
    const int64_t out_bstride = (int64_t)h->hop * T_max;
    size_t row = 0;
    for (const Step& st : steps) {
        rc = forward_impl(h, c, (int64_t)h->cf * T_max, T_max, out, out_bstride, out, out_bstride, st.n, st.frames,
                          plan_workspace(h, st.n, st.frames, workspace), stream, d_valid + row, 0, d_slots + row);
        if (rc != HIFICAR_OK) return rc;
        row += ((size_t)st.n + 1) & ~(size_t)1;
    }
    return HIFICAR_OK;
}

extern "C" int hificar_ar_loop(hificar_handle* h, const float* c, float* out, int B, int T_total, int chunk_frames,
                               void* workspace, size_t workspace_bytes, void* stream) {
    return hificar_ar_loop_ragged(h, c, nullptr, nullptr, out, B, T_total, chunk_frames, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// per-kernel event timing
// ------------------------------------------------------------------------------------------------
extern "C" int hificar_profile_begin(hificar_handle* h) {
    if (!h) return fail(HIFICAR_E_INVALID, "null handle");
    for (auto& r : h->prof) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    h->prof.clear();
    h->profiling = true;
    return HIFICAR_OK;
}

extern "C" int hificar_profile_end(hificar_handle* h, hificar_kernel_stat* stats, int max_stats, int* n_stats) {
    if (!h || !n_stats) return fail(HIFICAR_E_INVALID, "null argument");
    h->profiling = false;
    if (!h->prof.empty()) HIP_TRY(hipStreamSynchronize(h->prof_stream));
    std::vector<hificar_kernel_stat> agg;
    for (auto& r : h->prof) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r.e0, r.e1));
        size_t k = 0;
        for (; k < agg.size(); ++k)
            if (r.name == agg[k].name) break;
        if (k == agg.size()) {
            hificar_kernel_stat st;
            memset(&st, 0, sizeof(st));
            snprintf(st.name, sizeof(st.name), "%s", r.name.c_str());
            agg.push_back(st);
        }
        agg[k].launches += 1;
        agg[k].total_ms += ms;
        agg[k].flops += r.flops;
        agg[k].bytes += r.bytes;
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    h->prof.clear();
    std::sort(agg.begin(), agg.end(), [](const hificar_kernel_stat& a, const hificar_kernel_stat& b) { return a.total_ms > b.total_ms; });
    *n_stats = (int)agg.size();
    for (int i = 0; i < (int)agg.size() && i < max_stats && stats; ++i) stats[i] = agg[i];
    return HIFICAR_OK;
}

extern "C" int hificar_pcm16(const float* x, int16_t* y, size_t n, void* stream) {
    if (!x || !y) return fail(HIFICAR_E_INVALID, "hificar_pcm16: null pointer");
    if (n == 0) return HIFICAR_OK;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(pcm16_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n);
    HIP_TRY(hipGetLastError());
    return HIFICAR_OK;
}

#include "hificar_train.hip.inc"
#include "hificar_gblock.hip.inc"
#include "hificar_disc.hip.inc"
#include "hificar_mel.hip.inc"
