/*
 * hificar.h — C ABI of libhificar.so: the MI355X-native (gfx950) HiFi-GAN / HiFi-CAR generator
 * forward pass (200-Hz EMA/pitch frames -> 16-kHz waveform).
 *
 * The reference (articulatory/articulatory) has NO native boundary for this path: it is a Python
 * class protocol on top of PyTorch operators.  Each entry point below therefore names the
 * reference Python interface it stands in for (paths relative to the reference repo):
 *
 *   hificar_create          HiFiGANGenerator.__init__            articulatory/models/hifigan.py:24-196
 *   hificar_gblock_create   GBlockGenerator.__init__             articulatory/models/gblock_gen.py:17-109 (forward :111-132 = hificar_forward*)
 *   hificar_set_weight      load_state_dict + remove_weight_norm articulatory/utils/utils.py:340-342,
 *                                                                articulatory/models/hifigan.py:256-266
 *   hificar_finalize        model.eval().to(device)              egs/ema/voc1/local/predict_wav.py:114-115
 *   hificar_forward         HiFiGANGenerator.forward             articulatory/models/hifigan.py:198-239
 *   hificar_forward_cond    ... with spk_id= / ph= and the (out, ph_out) return of use_ph_loss    hifigan.py:212-220, 232-237
 *   hificar_ar_loop         ar_loop (non-WSOLA branch), batched  articulatory/bin/decode.py:31-83
 *   hificar_forward_ragged  the per-utterance loop over a dataset calling .inference()  egs/ema/voc1/local/predict_wav.py:124-137,
 *   hificar_ar_loop_ragged  ... or ar_loop(), one utterance at a time                    articulatory/bin/decode.py:292-351
 *                           (a batch of utterances of DIFFERENT lengths in one call; results per utterance are those of
 *                           the one-at-a-time loop, bit for bit)
 *   hificar_ar_loop_packed  the same dataset loop, continuously batched (a finished utterance's place is taken by the next)
 *   hificar_pcm16           sf.write(..., "PCM_16") sample conversion articulatory/bin/decode.py:319-324
 *   hificar_workspace_bytes (torch's caching allocator does this implicitly in the reference)
 *   hificar_last_error      Python exceptions / assert           articulatory/models/hifigan.py:78-80
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  Device pointers are HIP device pointers.
 *   - tensors at the boundary use the reference's layouts: features (B, C, T) fp32 with T contiguous,
 *     AR context (B, 1, ar_input) fp32, waveform (B, 1, hop*T) fp32.
 *   - every function returns 0 on success or a negative HIFICAR_E_* code; hificar_last_error() then
 *     returns a thread-local, human-readable message.  Nothing ever calls exit().
 *   - all device work is enqueued on the caller-supplied hipStream_t (passed as void*); no call
 *     synchronises the device except hificar_finalize() (one-time weight upload) and the FIRST call for a new
 *     (batch, frames) shape, which builds and uploads that shape's tile schedules (small device allocations + blocking
 *     copies; cached in the handle afterwards, so a warm-up call per shape keeps the steady state fully asynchronous).
 *   - a handle is not thread-safe; use one handle per (process, device) and, from one host thread at a time, preferably ONE stream
 *     per handle.  The tile-schedule arenas, the packed AR loop's step table and the caller's workspace are shared state of the
 *     handle: every entry point that enqueues work first makes its stream wait (hipStreamWaitEvent on an event recorded on the
 *     previous call's stream) for everything earlier calls on this handle enqueued, so calls on DIFFERENT streams are serialised
 *     behind each other, never overlapped.  Concurrency comes from separate handles (each with its own workspace), not from streams.
 */
#ifndef HIFICAR_H
#define HIFICAR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIFICAR_MAX_STAGES 8
#define HIFICAR_MAX_BLOCKS 4
#define HIFICAR_MAX_DILATIONS 4

#define HIFICAR_OK 0
#define HIFICAR_E_INVALID (-1)     /* bad argument / unsupported hyper-parameter            */
#define HIFICAR_E_STATE (-2)       /* call order (e.g. forward before finalize, missing weight) */
#define HIFICAR_E_HIP (-3)         /* a HIP runtime call failed (message carries hipGetErrorString) */
#define HIFICAR_E_WORKSPACE (-4)   /* workspace too small                                    */

/* arithmetic used by the convolution kernels */
#define HIFICAR_PREC_F32 0         /* v_mfma_f32_32x32x2_f32: exact fp32 multiply, fp32 accumulate */
#define HIFICAR_PREC_BF16X3 1      /* fp32 operands split hi+lo bf16, 3 bf16 MFMAs, fp32 accumulate */

typedef struct hificar_handle hificar_handle;

/* Mirrors the keyword arguments of HiFiGANGenerator.__init__ (hifigan.py:24-50) that affect
 * inference.  in_channels keeps the reference's meaning: feature dims + ar_output when use_ar. */
typedef struct hificar_config {
    int32_t in_channels;
    int32_t out_channels; /* must be 1 */
    int32_t channels;
    int32_t kernel_size;
    int32_t n_stages;
    int32_t upsample_scales[HIFICAR_MAX_STAGES];
    int32_t upsample_kernel_sizes[HIFICAR_MAX_STAGES];
    int32_t n_blocks;
    int32_t resblock_kernel_sizes[HIFICAR_MAX_BLOCKS];
    int32_t n_dilations[HIFICAR_MAX_BLOCKS];
    int32_t resblock_dilations[HIFICAR_MAX_BLOCKS][HIFICAR_MAX_DILATIONS];
    int32_t use_additional_convs; /* 0: a ResBlock layer is x + convs1[d](x) (residual_block.py:191-205, 217-221), no convs2 tensors */
    int32_t bias;                 /* ResBlock conv bias flag */
    float lrelu_slope;            /* nonlinear_activation_params.negative_slope */
    int32_t use_tanh;
    int32_t use_ar;
    int32_t ar_input;
    int32_t ar_hidden;
    int32_t ar_output;
    int32_t precision; /* HIFICAR_PREC_* */
    /* speaker / phoneme conditioning (hifigan.py:43-49, 176-189); all zero = off, as in every shipped YAML */
    int32_t use_spk_id;   /* adds spk_fc(spk_emb_mat[spk_id]) to every input channel of every frame (hifigan.py:212-216) */
    int32_t num_spk;
    int32_t spk_emb_size;
    int32_t use_ph;       /* appends ph_emb_mat[ph[b, t]] as ph_emb_size extra input channels (hifigan.py:217-220); in_channels counts them */
    int32_t num_ph;
    int32_t ph_emb_size;
    int32_t use_ph_loss;  /* second output: phoneme logits AvgPool1d(2*hop, hop, hop/2)(ph_fc(c)) at the frame rate (hifigan.py:232-237) */
} hificar_config;

/* Build an (empty) generator for these hyper-parameters on the current HIP device. */
int hificar_create(const hificar_config* cfg, hificar_handle** out);

/* The reference's OTHER a2w generator behind the same plugin surface: GBlockGenerator (articulatory/models/gblock_gen.py:14-132; GAN-TTS
 * style GBlocks, articulatory/layers/pytorch_layers.py:32-91).  Mirrors the keyword arguments of GBlockGenerator.__init__
 * (gblock_gen.py:17-31).  input conv (kernel_size) -> n_blocks GBlocks -> LeakyReLU(0.01) + conv (kernel_size) (+ tanh); GBlock i maps
 * channels / in_div[i] -> channels / out_div[i] channels with the reference's hard-coded plan in_div = 1,1,1,2,2,2,2,4,4,8,
 * out_div = 1,1,2,2,2,2,4,4,8,8 (gblock_gen.py:63-64) and upsamples by g_scales[i] (nearest neighbour):
 *     y = conv_k,d3(ReLU(conv_k(up(ReLU(x))))) + conv_1(up(x));   out = y + conv_k,d27(ReLU(conv_k,d9(ReLU(y))))
 * The handle is an ordinary hificar_handle: hificar_set_weight (names "input_conv.weight", "resamples.<i>.conv1.<1|2>.weight",
 * "resamples.<i>.conv1.<3|4>.weight", "resamples.<i>.res1.<0|1>.weight" (cout, cin, 1), "resamples.<i>.conv2.{1,3}.weight", the biases,
 * "output_conv.1.*", "ar_model.model.*", "spk_*" — the Sequential indices move by one when g_scales[i] > 1, as in the reference's
 * state_dict), hificar_finalize, hificar_forward[_cond / _ragged], hificar_ar_loop*, hificar_forward_train[_cond], hificar_backward[_cond],
 * hificar_set_parameters_device, ... all take it.  Exact-fp32 arithmetic only (res1 consumes the raw, un-activated rows).
 * Restrictions (each is a configuration the reference class itself cannot run, see oracle/make_golden_gblock.py): g_kernel_sizes odd;
 * the last GBlock must end at channels / 8 channels (n_blocks = 9 or 10). */
#define HIFICAR_MAX_GBLOCKS 10
typedef struct hificar_gblock_config {
    int32_t in_channels;  /* feature dims + ar_output when use_ar, as in hificar_config */
    int32_t out_channels; /* must be 1 */
    int32_t channels;
    int32_t kernel_size;  /* input / output conv */
    int32_t n_blocks;     /* len(g_scales) == len(g_kernel_sizes) */
    int32_t g_scales[HIFICAR_MAX_GBLOCKS];
    int32_t g_kernel_sizes[HIFICAR_MAX_GBLOCKS];
    int32_t use_tanh;
    int32_t use_ar;
    int32_t ar_input;
    int32_t ar_hidden;
    int32_t ar_output;
    int32_t use_spk_id;   /* gblock_gen.py:103-106, 123-127 */
    int32_t num_spk;
    int32_t spk_emb_size;
    int32_t precision;    /* must be HIFICAR_PREC_F32 */
} hificar_gblock_config;
int hificar_gblock_create(const hificar_gblock_config* cfg, hificar_handle** out);

/* Hand over one FOLDED tensor (weight-norm already baked: w = v*g/||v||) by its reference
 * state_dict name after remove_weight_norm(), e.g. "input_conv.weight", "upsamples.0.1.weight"
 * (Cin,Cout,K), "blocks.3.convs1.2.1.bias", "output_conv.1.weight", "ar_model.model.4.weight".
 * `data` is a HOST pointer to contiguous fp32 in the reference's layout; the library repacks into
 * its kernel layout and the caller keeps ownership. */
int hificar_set_weight(hificar_handle* h, const char* name, const float* data, const int64_t* shape, int ndim);

/* Check that every tensor arrived, repack, upload to the device.  Synchronises the device once. */
int hificar_finalize(hificar_handle* h);

/* Switch the conv arithmetic after creation (HIFICAR_PREC_*).  Cheap; weights for both modes are resident. */
int hificar_set_precision(hificar_handle* h, int precision);

/* Bytes of device scratch hificar_forward / hificar_ar_loop need for B utterances of T frames per call
 * (for hificar_ar_loop pass T = chunk_frames). */
size_t hificar_workspace_bytes(const hificar_handle* h, int B, int T);

/* One generator forward.  c: (B, in_channels - ar_output*use_ar, T) contiguous device fp32;
 * ar: (B, 1, ar_input) contiguous device fp32, or NULL when !use_ar; out: (B, 1, hop*T) device fp32,
 * hop = prod(upsample_scales).  Inputs are not modified. */
int hificar_forward(hificar_handle* h, const float* c, const float* ar, float* out, int B, int T,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Batched autoregressive synthesis of B equal-length utterances (decode.py:54-83 per utterance):
 * c: (B, C, T_total) device fp32; out: (B, hop*T_total) device fp32.  Chunks of chunk_frames frames
 * (= batch_max_steps / hop_size, decode.py:50) run sequentially, each conditioned on the last ar_input
 * output samples of the previous one (zeros for the first); the last chunk may be shorter.
 * Requires ar_input <= hop*chunk_frames (the only case in which the reference's loop is well formed). */
int hificar_ar_loop(hificar_handle* h, const float* c, float* out, int B, int T_total, int chunk_frames,
                    void* workspace, size_t workspace_bytes, void* stream);

/* hificar_forward with the conditioning inputs of HiFiGANGenerator.forward(c, spk_id=, ar=, ph=) (hifigan.py:198-239):
 * spk_id: device pointer to B int32 speaker indices (use_spk_id) or NULL; ph: device pointer to (B, T) int32 phoneme indices
 * (use_ph) or NULL; ph_out: device pointer to (B, num_ph, T) fp32 (use_ph_loss: the reference then returns (out, ph_out)) or NULL.
 * lengths as in hificar_forward_ragged (NULL: all T).  Features c: (B, in_channels - ar_output*use_ar - ph_emb_size*use_ph, T). */
int hificar_forward_cond(hificar_handle* h, const float* c, const float* ar, const int32_t* spk_id, const int32_t* ph,
                         const int32_t* lengths, float* out, float* ph_out, int B, int T, void* workspace, size_t workspace_bytes,
                         void* stream);

/* Ragged batches: B utterances of different lengths, padded to a common T (T_total) in `c`.
 * lengths: DEVICE pointer to B int32 frame counts (0 <= lengths[b] <= T), or NULL (all T).  Utterance b is
 * synthesised exactly as if it were alone: frames >= lengths[b] do not exist for it (its convs see zero padding
 * there — also in the shorter last AR chunk, decode.py:56-58 — and nothing is written to out[b, hop*lengths[b]:],
 * which keeps whatever the caller put there).  This is the dataset loop of predict_wav.py:124-137 /
 * decode.py:292-351 run B utterances at a time. */
int hificar_forward_ragged(hificar_handle* h, const float* c, const float* ar, const int32_t* lengths, float* out, int B,
                           int T, void* workspace, size_t workspace_bytes, void* stream);
int hificar_ar_loop_ragged(hificar_handle* h, const float* c, const int32_t* lengths, const int32_t* lengths_host, float* out,
                           int B, int T_total, int chunk_frames, void* workspace, size_t workspace_bytes, void* stream);
/* lengths_host: optional HOST copy of the same B values (NULL = unknown to the host).  With it, every AR step is launched
 * only over the utterances still running — the batch prefix up to the last one longer than the step's first frame, i.e.
 * all of them and nothing else when the batch is sorted longest first — instead of masking finished ones on the device. */

/* Packed ("continuously batched") AR synthesis of a whole list of utterances: c (N, C, T_max) and out (N, hop*T_max) on the
 * device, lengths_host N frame counts on the HOST.  At most `batch` utterances are in flight; when one finishes, the next
 * one of the list takes its place in the following step, so all steps but the last few run a full batch whatever the
 * length distribution (list the utterances longest first for the shortest tail).  Per utterance the result is that of
 * hificar_ar_loop on it alone, bit for bit; out[u, hop*lengths[u]:] is left untouched.
 * workspace: hificar_workspace_bytes(h, batch, chunk_frames).  Stands in for the dataset loop of
 * articulatory/bin/decode.py:292-351 / egs/ema/voc1/local/predict_wav.py:124-137. */
int hificar_ar_loop_packed(hificar_handle* h, const float* c, const int32_t* lengths_host, float* out, int N, int T_max,
                           int chunk_frames, int batch, void* workspace, size_t workspace_bytes, void* stream);

/* float waveform in [-1, 1] -> 16-bit PCM on the device: y = clip(round_half_even(x *_f32 32767.f), -32768, 32767) — the product in
 * float32, as libsndfile's f2s_array computes it (lrintf(src * 32767.f)); libsndfile wraps outside [-1, 1] where this clips.
 * What the reference's sf.write(..., "PCM_16") does on the host after the device->host copy
 * (articulatory/bin/decode.py:319-324); doing it before the copy / the multi-GPU gather halves the bytes moved.
 * x, y: device pointers, n elements. */
int hificar_pcm16(const float* x, int16_t* y, size_t n, void* stream);

/* Algorithmic multiply-accumulates of one forward of B x T frames (conv + MLP MACs; bias/activation
 * excluded) — the constant SURVEY.md §8(d) defines; used by bench.py for the roofline figure. */
double hificar_macs(const hificar_handle* h, int B, int T);

/* Per-kernel timing with HIP events on the launch stream (bench.py's roofline leg; no reference
 * counterpart — the reference times whole utterances with time.time(), articulatory/bin/decode.py:302-318).
 * Between hificar_profile_begin and hificar_profile_end every kernel launch of this handle is bracketed
 * by two hipEvents.  hificar_profile_end synchronises the stream used, then fills up to `max_stats`
 * entries (one per distinct kernel) and writes the number of distinct kernels to *n_stats. */
typedef struct hificar_kernel_stat {
    char name[96];     /* e.g. "conv_bf16x3_kernel<4,1,4,4>" (template args as in the rocprof kernel name) */
    int64_t launches;
    double total_ms;   /* sum of hipEventElapsedTime over the launches */
    double flops;      /* algorithmic FLOPs (2 x MACs) of those launches */
    double bytes;      /* algorithmic bytes (inputs + outputs + weights read once) of those launches */
} hificar_kernel_stat;
int hificar_profile_begin(hificar_handle* h);
int hificar_profile_end(hificar_handle* h, hificar_kernel_stat* stats, int max_stats, int* n_stats);

/* ---------------------------------------------------------------------------------------------------------------------------
 * Training: the generator half of the reference's train step (articulatory/bin/train.py:241-440: y_ = generator(x, ar=ar) under
 * autograd, gen_loss.backward()).  The reference has no native boundary here either (PyTorch autograd through torch.nn modules);
 * these entry points stand in for  HiFiGANGenerator.forward  in training mode and the autograd graph behind it.
 * Exact-fp32 arithmetic only.
 *
 *   hificar_set_weight_device  one FOLDED parameter (name / layout as hificar_set_weight) from DEVICE memory, repacked on the
 *                              device on `stream` — the weights of a model in training live on the device and change every step
 *                              (the weight-norm fold w = v * g / ||v|| and its gradient stay with the caller's autograd).
 *                              After the first call the handle serves fp32 layer-by-layer kernels only.
 *   hificar_set_parameters_device  EVERY parameter in its RAW state_dict form from device memory in one call (two launches):
 *                              "<conv>.weight_g" + "<conv>.weight_v" for a weight-normed conv (torch.nn.utils.weight_norm, dim 0 —
 *                              hifigan.py:268-278), "<conv>.weight" for a plain one, biases, the PastFCEncoder tensors.  The fold
 *                              w = g v / ||v|| runs on the device into a master copy every pack is refreshed from.
 *   hificar_weight_norm_backward  hificar_backward's folded gradients -> gradients of those raw parameters (dg, dv of the weight
 *                              norm; the others copied): `raw_grads` holds hificar_raw_grad_floats(h) floats, one slot per entry of
 *                              the last hificar_set_parameters_device call, in its order, each slot rounded up to 4 floats.
 *   hificar_forward_train      hificar_forward that also keeps every activation the backward pass needs in `tape`
 *                              (hificar_tape_bytes(h, B, T) bytes, 256-byte aligned, caller-owned until hificar_backward returns)
 *   hificar_backward           gradients of one hificar_forward_train: dout (B, hop*T) gradient of the waveform, out the waveform
 *                              the forward returned -> `grads` (hificar_grad_floats(h) floats; parameter i of hificar_grad_count(h)
 *                              at the offset hificar_grad_info reports, in the reference's folded layout), dc (B, C, T) gradient of
 *                              the features or NULL, dar (B, ar_input) gradient of the AR context or NULL.
 *                              workspace: hificar_backward_workspace_bytes(h, B, T) bytes.
 *   hificar_forward_train_cond / hificar_backward_cond   the same pair for the conditioned generator (train.py:276 passes spk_id= / ph=
 *                              under autograd; hifigan.py:176-189, 212-220, 232-237): spk_id (B) / ph (B, T) int32 as in
 *                              hificar_forward_cond, ph_out (B, num_ph, T) the phoneme-loss head's output (use_ph_loss) or NULL; the
 *                              backward takes the SAME spk_id / ph again and dph_out (B, num_ph, T), the gradient of ph_out (NULL: ph_out
 *                              took no part in the loss).  `grads` then also holds spk_emb_mat.weight, spk_fc.{weight,bias},
 *                              ph_emb_mat.weight, ph_fc.{weight,bias} (hificar_grad_info); hificar_set_parameters_device takes them too.
 * --------------------------------------------------------------------------------------------------------------------------- */
int hificar_set_weight_device(hificar_handle* h, const char* name, const float* data, void* stream);
int hificar_set_parameters_device(hificar_handle* h, const char* const* names, const float* const* data, int n, void* stream);
int64_t hificar_raw_grad_floats(const hificar_handle* h);
int hificar_weight_norm_backward(hificar_handle* h, const float* grads, float* raw_grads, void* stream);
size_t hificar_tape_bytes(const hificar_handle* h, int B, int T);
int hificar_forward_train(hificar_handle* h, const float* c, const float* ar, float* out, int B, int T, void* workspace,
                          size_t workspace_bytes, void* tape, size_t tape_bytes, void* stream);
size_t hificar_backward_workspace_bytes(const hificar_handle* h, int B, int T);
int hificar_grad_count(hificar_handle* h);
int hificar_grad_info(hificar_handle* h, int i, char* name96, int64_t* offset, int64_t* numel);
int64_t hificar_grad_floats(hificar_handle* h);
int hificar_backward(hificar_handle* h, const float* dout, const float* out, int B, int T, const void* tape, size_t tape_bytes,
                     float* grads, float* dc, float* dar, void* workspace, size_t workspace_bytes, void* stream);
/* Data-parallel training: gradient BUCKETS.  The reference meant to wrap both networks in DistributedDataParallel (articulatory/bin/
 * train.py:1790-1801), which all-reduces buckets of gradients while the backward pass still runs.  Here a bucket is a group of parameters
 * whose gradients are complete at a known point of hificar_backward(_cond) / hificar_disc_backward:
 *   generator      bucket b < n_stages: the ResBlocks of stage n_stages - 1 - b (bucket 0 also the output conv and the phoneme head);
 *                  bucket n_stages: everything else (input conv, upsamplers, PastFCEncoder, conditioning tensors) — hificar_grad_bucket_count
 *   discriminator  one bucket per sub-discriminator (scales first, then periods) — hificar_disc_grad_bucket_count
 * hificar_raw_param_bucket(h, i) is the bucket of raw parameter i of the last hificar_set_parameters_device call.  With a callback set,
 * the backward calls fn(bucket, stream, user) on the host right after the bucket's last gradient kernel has been enqueued on `stream`
 * (the call's stream, or the sub-discriminator's side stream): the callee runs hificar_weight_norm_backward_bucket on that stream and
 * starts its collective behind it, while the backward goes on enqueueing.  The bucket variants add no cross-stream ordering. */
typedef void (*hificar_bucket_fn)(int bucket, void* stream, void* user);
int hificar_grad_bucket_count(hificar_handle* h);
int hificar_raw_param_bucket(const hificar_handle* h, int i);
int hificar_set_bucket_callback(hificar_handle* h, hificar_bucket_fn fn, void* user);
int hificar_weight_norm_backward_bucket(hificar_handle* h, const float* grads, float* raw_grads, int bucket, void* stream);
int hificar_forward_train_cond(hificar_handle* h, const float* c, const float* ar, const int32_t* spk_id, const int32_t* ph, float* out,
                               float* ph_out, int B, int T, void* workspace, size_t workspace_bytes, void* tape, size_t tape_bytes, void* stream);
int hificar_backward_cond(hificar_handle* h, const float* dout, const float* dph_out, const float* out, const int32_t* spk_id,
                          const int32_t* ph, int B, int T, const void* tape, size_t tape_bytes, float* grads, float* dc, float* dar,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Parity aid: per-layer intermediates of the forwards that follow, copied into caller buffers in the reference's (B, C, L)
 * layout — what a forward hook on the reference's modules (articulatory/models/hifigan.py:221-231,
 * articulatory/layers/residual_block.py:217-221) returns.  Names:
 *   "ar_feats"                (B, ar_output)            PastFCEncoder output (pytorch_layers.py:459-460)
 *   "input_conv"              (B, channels, T)          hifigan.py:221
 *   "upsamples.<i>"           (B, C_i, L_i)             LeakyReLU + ConvTranspose1d output, hifigan.py:224
 *   "blocks.<n>.convs1.<d>"   (B, C_i, L_i)             residual_block.py:218 (runs that layer pair unfused)
 *   "blocks.<n>.x.<d>"        (B, C_i, L_i)             residual stream after dilation d: xt + x, residual_block.py:221
 *   "blocks.<n>"              (B, C_i, L_i)             block output, hifigan.py:228
 * GBlockGenerator handles (hooks on articulatory/layers/pytorch_layers.py:85-91):
 *   "ar_feats", "input_conv"  as above
 *   "resamples.<i>.conv1a"    (B, Cout_i, L_i)          first conv of conv1 (pre-activation)
 *   "resamples.<i>.res1"      (B, Cout_i, L_i)          res1(x)
 *   "resamples.<i>.mid"       (B, Cout_i, L_i)          conv1(x) + res1(x)
 *   "resamples.<i>"           (B, Cout_i, L_i)          GBlock output
 * dst: device pointer to `capacity` floats (an error is returned by the forward if it is too small); dst = NULL removes the
 * tap, name = NULL removes all.  Taps add copies and (for convs1) extra launches: not for timed runs. */
int hificar_debug_tap(hificar_handle* h, const char* name, float* dst, size_t capacity);

void hificar_destroy(hificar_handle* h);

const char* hificar_last_error(void);

/* Library / build identification, e.g. "hificar 0.1 gfx950". */
const char* hificar_version(void);

/* ---------------------------------------------------------------------------------------------------------------------------
 * Discriminators of the train step: HiFiGANMultiScaleMultiPeriodDiscriminator (articulatory/models/hifigan.py:741-825; the
 * scale discriminators :503-643, the period discriminators :317-417), called at articulatory/bin/train.py:341-347 (generator
 * step: D(fake) with the gradient flowing back into the generator's waveform, D(real) without) and :421-424 (discriminator
 * step).  The reference has no native boundary here (torch.nn modules under autograd); these entry points stand in for
 * `discriminator(x)` and the autograd graph behind it.  Exact fp32.
 *
 * Layers are described explicitly (the host side derives them from discriminator_params exactly as the reference's constructors
 * do): s_* = the Conv1d stack of ONE scale discriminator (the last layer has no activation), p_* = the Conv2d (k, 1) stack of
 * ONE period discriminator (last = output_conv).  Parameter names follow the reference state_dict
 * ("msd.discriminators.0.layers.1.0.weight", "mpd.discriminators.2.convs.0.0.weight_g", ...).
 *
 *   hificar_disc_set_parameters_device   every RAW parameter from device memory (as hificar_set_parameters_device)
 *   hificar_disc_forward                 x (B, 1, T) -> every layer output of every sub-discriminator, kept in `tape`
 *                                        (hificar_disc_tape_bytes, 256-byte aligned, caller-owned); output buffer i =
 *                                        one (sub-discriminator, layer, group): [nseq][rows][pitch] floats, `channels` valid
 *                                        columns; scale discriminators: nseq = B, rows = time; period discriminators:
 *                                        nseq = B * period (sequence b * period + column), rows = T' / period.  Groups of a
 *                                        grouped conv are separate buffers (channel c of the layer = group c / channels).
 *   hificar_disc_backward                douts[i] = gradient of output buffer i (same layout) or NULL; -> grads (folded
 *                                        parameters, hificar_disc_grad_floats, offsets from hificar_disc_param_info) or NULL,
 *                                        dx (B, T) or NULL
 *   hificar_disc_weight_norm_backward    folded gradients -> raw parameter gradients (as hificar_weight_norm_backward)
 *   hificar_disc_engine                  the engine handle, for hificar_profile_begin / hificar_profile_end
 * --------------------------------------------------------------------------------------------------------------------------- */
#define HIFICAR_DISC_MAX_SUBS 8
#define HIFICAR_DISC_MAX_LAYERS 12
typedef struct hificar_disc_config {
    int n_scales;                 /* scale discriminators (hifigan.py:666-738), AvgPool1d between them */
    int pool_kernel, pool_stride, pool_pad;
    int s_n_layers;
    int s_cin[HIFICAR_DISC_MAX_LAYERS], s_cout[HIFICAR_DISC_MAX_LAYERS], s_k[HIFICAR_DISC_MAX_LAYERS], s_stride[HIFICAR_DISC_MAX_LAYERS],
        s_pad[HIFICAR_DISC_MAX_LAYERS], s_groups[HIFICAR_DISC_MAX_LAYERS];
    int s_bias;
    float s_slope;
    int n_periods;                /* period discriminators (hifigan.py:451-500) */
    int periods[HIFICAR_DISC_MAX_SUBS];
    int p_n_layers;
    int p_cin[HIFICAR_DISC_MAX_LAYERS], p_cout[HIFICAR_DISC_MAX_LAYERS], p_k[HIFICAR_DISC_MAX_LAYERS], p_stride[HIFICAR_DISC_MAX_LAYERS],
        p_pad[HIFICAR_DISC_MAX_LAYERS];
    float p_slope;
} hificar_disc_config;

typedef struct hificar_disc_output {
    int sub, layer, group, n_groups;
    int period;                   /* 0: scale discriminator */
    int64_t offset_bytes;         /* inside the tape */
    int nseq, rows, pitch, channels;
} hificar_disc_output;

/* GAN criterion on the forward tapes (articulatory/losses/adversarial_loss.py:12-123, feat_match_loss.py:12-54 as combined at
 * train.py:341-362 / :421-424).  hificar_disc_loss: mode 0 = generator side (adversarial loss of the fake pass's final outputs +
 * feature matching against the real pass `tape_ref`, NULL: none): values = {adv, fm, lambda_adv * (adv + lambda_feat_match * fm)};
 * mode 1 / 2 = discriminator side, fake / real pass: values = {loss, 0, loss}.  values: 3 device floats; douts:
 * hificar_disc_dout_floats floats = d(values[2]) / d(every output buffer), consumed by hificar_disc_backward_flat. */
typedef struct hificar_gan_loss_config {
    int loss_type;                 /* 0 mse, 1 hinge */
    int average_by_discriminators; /* adversarial losses */
    int fm_average_by_layers, fm_average_by_discriminators, fm_include_final_outputs;
    float lambda_adv, lambda_feat_match;
} hificar_gan_loss_config;

typedef struct hificar_disc hificar_disc;
size_t hificar_disc_dout_floats(const hificar_disc* d, int B, int T);
int hificar_disc_loss(hificar_disc* d, const hificar_gan_loss_config* cfg, int mode, const void* tape, const void* tape_ref, int B, int T,
                      float* values3, float* douts, void* stream);
int hificar_disc_backward_flat(hificar_disc* d, const float* douts, int mode, int with_fm, int fm_include_final_outputs, int B, int T,
                               const void* tape, size_t tape_bytes, float* grads, float* dx, void* workspace, size_t workspace_bytes,
                               void* stream);
int hificar_disc_create(const hificar_disc_config* cfg, hificar_disc** out);
void hificar_disc_destroy(hificar_disc* d);
hificar_handle* hificar_disc_engine(hificar_disc* d);
int hificar_disc_param_count(const hificar_disc* d);
int hificar_disc_param_info(const hificar_disc* d, int i, char* name96, int64_t* shape4, int* ndim, int64_t* offset);
int64_t hificar_disc_grad_floats(const hificar_disc* d);
int64_t hificar_disc_raw_grad_floats(const hificar_disc* d);
int hificar_disc_set_parameters_device(hificar_disc* d, const char* const* names, const float* const* data, int n, void* stream);
int hificar_disc_weight_norm_backward(hificar_disc* d, const float* grads, float* raw_grads, void* stream);
size_t hificar_disc_tape_bytes(const hificar_disc* d, int B, int T);
/* Algorithmic multiply-accumulates of one discriminator forward over (B, 1, T) (every Conv1d / Conv2d of hifigan.py:317-825: output
 * positions x cout x cin / groups x k) — the training roofline's work unit, as hificar_macs is the generator's. */
double hificar_disc_macs(const hificar_disc* d, int B, int T);
/* gradient buckets of the discriminators (see hificar_bucket_fn above): one per sub-discriminator */
int hificar_disc_grad_bucket_count(const hificar_disc* d);
int hificar_disc_raw_param_bucket(const hificar_disc* d, int i);
int hificar_disc_bucket_folded_range(const hificar_disc* d, int bucket, int64_t* offset, int64_t* numel);
int hificar_disc_set_bucket_callback(hificar_disc* d, hificar_bucket_fn fn, void* user);
int hificar_disc_weight_norm_backward_bucket(hificar_disc* d, const float* grads, float* raw_grads, int bucket, void* stream);
/* The discriminator update backpropagates two passes (real, fake: train.py:405-437 sums their losses).  on != 0: the following
 * hificar_disc_backward* calls ADD their parameter gradients to what `grads` holds instead of overwriting it, so the second pass lands in
 * the first pass's buffer (no second buffer, no add pass).  scale_device: device scalar (or NULL) that the following
 * hificar_disc_weight_norm_backward[_bucket] calls multiply every raw gradient by while writing it — autograd's upstream gradient of the
 * scalar loss.  Both are per-handle switches; the caller resets them (0 / NULL) after the step. */
int hificar_disc_set_grad_accumulate(hificar_disc* d, int on);
int hificar_disc_set_grad_scale(hificar_disc* d, const float* scale_device);
size_t hificar_disc_backward_workspace_bytes(const hificar_disc* d, int B, int T);
int hificar_disc_output_count(const hificar_disc* d);
int hificar_disc_output_info(const hificar_disc* d, int B, int T, int i, hificar_disc_output* out);
int hificar_disc_forward(hificar_disc* d, const float* x, int B, int T, void* tape, size_t tape_bytes, void* stream);
int hificar_disc_backward(hificar_disc* d, const float* const* douts, int B, int T, const void* tape, size_t tape_bytes, float* grads,
                          float* dx, void* workspace, size_t workspace_bytes, void* stream);

/* Mel-spectrogram loss (articulatory/losses/mel_loss.py:114-166; train.py:308-311): F.l1_loss(log mel(y_hat), log mel(y)) and its
 * gradient with respect to y_hat in one call.  melmat: (num_mels, fft_size / 2 + 1) filterbank, host memory (librosa.filters.mel in the
 * reference, mel_loss.py:56-62).  window: hann (periodic, torch.hann_window), center = True, normalized = False, onesided = True.
 * log_base: 0 natural, 2 or 10. */
typedef struct hificar_mel_config {
    int fft_size, hop_size, win_length, num_mels;
    float eps;
    int log_base;
    int mode;   /* 0: mel-spectrogram loss; 1: one resolution of the multi-resolution STFT loss (melmat, num_mels, log_base unused) */
} hificar_mel_config;
typedef struct hificar_mel hificar_mel;
int hificar_mel_create(const hificar_mel_config* cfg, const float* melmat, hificar_mel** out);
void hificar_mel_destroy(hificar_mel* m);
size_t hificar_mel_workspace_bytes(const hificar_mel* m, int B, int T);
/* One resolution of MultiResolutionSTFTLoss (articulatory/losses/stft_loss.py:87-125; train.py:289-290) on a mode-1 handle: forward ->
 * values = {spectral convergence, log STFT magnitude} (2 device floats), the magnitudes stay in `workspace`; backward (same workspace):
 * dy_hat (B, T) = d(gweights[0] * sc + gweights[1] * mag) / dy_hat, gweights = 2 DEVICE floats (the upstream gradients). */
int hificar_stft_loss_forward(hificar_mel* m, const float* y_hat, const float* y, int B, int T, float* values2, void* workspace,
                              size_t workspace_bytes, void* stream);
int hificar_stft_loss_backward(hificar_mel* m, int B, int T, const float* gweights2, float* dy_hat, void* workspace, size_t workspace_bytes,
                               void* stream);
int hificar_mel_loss(hificar_mel* m, const float* y_hat, const float* y, int B, int T, float* value, float* dy_hat, void* workspace,
                     size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HIFICAR_H */
