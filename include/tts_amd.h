/* tts_amd — C ABI of the MI355X-native (gfx950) VITS / Glow-TTS / HiFiGAN inference hot path.
 *
 * Every entry point is `extern "C"`, takes plain device pointers + sizes (no torch types) and
 * a `hipStream_t` passed as `void*` (NULL = the null stream).  All pointers are DEVICE pointers
 * unless the comment says "host".  The caller owns every buffer; the library owns nothing but
 * kernels (and, for the *_c mirror entry, a stream-ordered scratch allocation).
 * Return value: 0 on success, negative TTSAMD_ERR_* otherwise; never throws/aborts across the
 * ABI.  `ttsamd_last_error()` returns a thread-local message for the last failure.
 *
 * Each declaration cites the reference interface (coqui-ai/TTS v0.22.0, paths relative to the
 * reference root) that it replaces.  Layout everywhere: fp32, channels-first `[B, C, T]`
 * contiguous, exactly like the reference's tensors.
 */
#ifndef TTS_AMD_H
#define TTS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTSAMD_OK 0
#define TTSAMD_ERR_INVALID (-1)     /* bad argument (shape, NULL, unsupported kernel size ...) */
#define TTSAMD_ERR_UNSUPPORTED (-2) /* valid request outside what the kernels cover (documented limits) */
#define TTSAMD_ERR_HIP (-3)         /* a HIP runtime call / launch failed */
#define TTSAMD_ERR_INTERNAL (-4)    /* host-side failure inside the library (out of host memory, an unexpected C++ exception): the
                                       handle entries run behind an exception barrier, nothing unwinds into the caller */

const char *ttsamd_last_error(void);
/* ABI version (bumped on any signature change) and the gfx arch the library was compiled for. */
int ttsamd_abi_version(void);
/* Kernel launches issued through this library since it was loaded (all threads; launches recorded into a stream capture
 * count when they are recorded, not when the graph replays): "launches per request" of a chain = the difference around one
 * eager run of it. */
uint64_t ttsamd_launch_count(void);
const char *ttsamd_arch(void);

/* Streams owned by the caller through the library: a dedicated HIP stream on the current device (priority 0 = normal,
 * -1 = high, 1 = low; clamped to the device's range).  The host side uses these for request lanes and for the MRF branch
 * streams of a HiFiGAN generator, so that no two of them can ever be the same underlying stream (a framework stream pool
 * hands its streams out round-robin and re-uses them).  replaces: the `torch.cuda.Stream()` objects around
 * TTS/utils/synthesizer.py:384's sentence loop a torch caller would create. */
int ttsamd_stream_create(int priority, void **stream_out);
int ttsamd_stream_destroy(void *stream);

/* ------------------------------------------------------------------------------------------
 * Monotonic alignment search
 * replaces: TTS/tts/utils/monotonic_align/core.pyx:42-47  maximum_path_c(paths, values, t_xs,
 *           t_ys, max_neg_val=-1e9)   (and maximum_path_each, core.pyx:11-37)
 * ---------------------------------------------------------------------------------------- */

/* Exact mirror of `maximum_path_c`: `values` fp32 [b,t_x,t_y] is updated IN PLACE inside the DP
 * band, `paths` int32 [b,t_x,t_y] must be pre-zeroed by the caller (helpers.py:188) and receives
 * 1 on the chosen path.  t_xs/t_ys int32 [b] (device).  Bit-exact with the reference for every
 * valid input (1 <= t_xs[i] <= t_ys[i]); items with t_x<=0 or t_y<=0 are left untouched.
 * No limit on t_x or t_y (core.pyx has none): up to 512 rows a skewed pipeline of DP waves, up to 2048 one DP wave with
 * the column in registers, beyond that a column-stepping workgroup with its column state in LDS (<= 16 384 rows) or in
 * the workspace. */
int ttsamd_maximum_path_c(int32_t *paths, float *values, const int32_t *t_xs, const int32_t *t_ys,
                          int b, int t_x, int t_y, float max_neg_val, void *stream);

/* Scratch bytes needed by ttsamd_maximum_path (direction bit-planes; + two columns of state per item for t_x > 16 384). */
size_t ttsamd_maximum_path_workspace_bytes(int b, int t_x, int t_y);

/* Fused form of helpers.maximum_path_cython (TTS/tts/utils/helpers.py:178-194):
 *   value = value * mask (mask may be NULL = all ones; applied on the fly, `values_in` is const),
 *   DP + backtrack; `paths` is fully written (zeros included) unless TTSAMD_MAS_PATHS_PREZEROED.
 *   `dp_values_out` (may be NULL, may alias values_in) receives the in-place-updated values.
 *   `paths_f32` selects a float32 0/1 output instead of int32 (the reference casts the int path
 *   back to value.dtype, helpers.py:194). */
#define TTSAMD_MAS_PATHS_PREZEROED 1
#define TTSAMD_MAS_PATHS_F32 2
int ttsamd_maximum_path(void *paths, const float *values_in, const float *mask, float *dp_values_out,
                        const int32_t *t_xs, const int32_t *t_ys, int b, int t_x, int t_y,
                        float max_neg_val, void *workspace, size_t workspace_bytes, int flags,
                        void *stream);

/* Log-likelihood matrix feeding MAS — replaces the einsum/matmul block of Vits.forward_mas (TTS/tts/models/vits.py:912-918;
 * glow_order=0: logp2+logp3+logp1+logp4) and of GlowTTS.forward / inference_with_MAS (TTS/tts/models/glow_tts.py:241-247,
 * 291-296; glow_order=1: logp1+logp2+logp3+logp4).  z [B,C,T_y] (latent), m / logs [B,C,T_x] (text-side prior stats)
 * -> logp [B,T_x,T_y], ready for ttsamd_maximum_path on the device. */
int ttsamd_mas_logp(float *logp, const float *z, const float *m, const float *logs, int batch, int c, int t_x, int t_y,
                    int glow_order, void *stream);

/* t_xs[i] = sum_x mask[i,x,0], t_ys[i] = sum_y mask[i,0,y]   (helpers.py:191-192). */
int ttsamd_mask_lengths(int32_t *t_xs, int32_t *t_ys, const float *mask, int b, int t_x, int t_y,
                        void *stream);


/* ------------------------------------------------------------------------------------------
 * Conv1d / ConvTranspose1d as implicit GEMM on the fp32-input MFMA (v_mfma_f32_32x32x2_f32: exact
 * fp32 == an fmaf chain, at the fp32 vector peak rate) with fused prologue/epilogue.
 * replaces (torch ops behind): torch.nn.Conv1d / ConvTranspose1d + F.leaky_relu + residual adds in
 *   TTS/vocoder/models/hifigan_generator.py:83-98 (ResBlock1), :150-155 (ResBlock2), :236-265
 *   (HifiganGenerator.forward); TTS/tts/layers/generic/wavenet.py:92-116 (WN.forward, incl. the
 *   TorchScript fused_add_tanh_sigmoid_multiply :6-13); TTS/tts/layers/vits/networks.py:138-166
 *   (ResidualCouplingBlock); TTS/tts/layers/glow_tts/transformer.py:106-114,290-295 (q/k/v/o + FFN).
 * ---------------------------------------------------------------------------------------- */
#define TTSAMD_ACT_NONE 0
#define TTSAMD_ACT_LRELU 1 /* input side: leaky_relu(x, in_slope) */
#define TTSAMD_ACT_RELU 1  /* output side */
#define TTSAMD_ACT_TANH 2  /* output side */
#define TTSAMD_ACT_GELU 3  /* norm kernel only: exact erf GELU (F.gelu default) */

#define TTSAMD_CONV_NORMAL 0
/* WN gate (ABI v2 row order): packed 32-row tile m = tanh channels [16m, 16m+16) then the matching sigmoid channels;
 * writes tanh(.)*sigmoid(.) to out channel 16m+i (wavenet.py:6-13). c_out = number of packed rows (2H, H % 16 == 0 or the
 * last tile zero-padded per half). */
#define TTSAMD_CONV_GATE 1
/* ConvTranspose1d (kernel 2u, stride u, pad u/2) in polyphase form: packed row m = co*u + r holds
 * the 2-tap filter of phase r; column q is written to y[b, co, q*u + r - shuffle_pad]. */
#define TTSAMD_CONV_SHUFFLE 2
/* affine coupling, mean only (networks.py:164): y = (res - (conv+bias)*mask) * mask */
#define TTSAMD_CONV_COUPLE 3
/* WN res/skip 1x1 conv (wavenet.py:109-114): rows < split_row: y[row] = (res[row] + v) * out_mask;
 * rows >= split_row: y2[row-split_row] = accum[row-split_row] + v (accum may be NULL: first layer). */
#define TTSAMD_CONV_RES_SKIP 4
/* Glow affine coupling (glow.py:216-224): packed rows paired like GATE: tile m = t rows of channels [16m, 16m+16) then their
 * s rows; y[16m+i] = (res[16m+i] - t) * exp(-s) * out_mask.  c_out = packed rows (32 per tile; split_row = coupled channels). */
#define TTSAMD_CONV_COUPLE_AFFINE 5
/* forward direction of the same coupling (glow.py:225): y[16m+i] = (t + exp(s) * res[16m+i]) * out_mask */
#define TTSAMD_CONV_COUPLE_AFFINE_FWD 6
/* COUPLE_AFFINE followed, in the same epilogue, by the rest of the reverse flow block (glow.py:107-137, normalization.py:98-103):
 * InvConvNear^-1 (the stored 4x4 inverse mixes channels {2i, 2i+1, C/2 + 2i, C/2 + 2i + 1}) and ActNorm^-1, IN PLACE on the whole
 * [2 * split_row] channel tensor: `res` / `y` point at its coupled half (rows split_row..), the other half sits split_row rows
 * before them.  `y2` points at the block's parameters on the device: 16 floats (4x4 inverse, row major), then bias[C], logs[C].
 * Operation for operation what COUPLE_AFFINE + ttsamd_glow_invconv_actnorm(forward = 0) compute: bitwise the same result. */
#define TTSAMD_CONV_COUPLE_AFFINE_MIX 7

typedef struct ttsamd_conv1d_args {
    const float *x;        /* x[b,ci,t] = x[b*x_bstride + ci*x_rstride + t], t in [0,t_in) */
    int64_t x_bstride, x_rstride;
    int32_t c_in, t_in;
    const float *w_packed; /* from ttsamd_conv1d_pack_weights */
    const float *bias;     /* [c_out] in packed-row order, or NULL */
    int32_t c_out, kernel, dilation, pad_left;
    float *y;              /* y[b,co,t] = y[b*y_bstride + co*y_rstride + t] */
    int64_t y_bstride, y_rstride;
    int32_t t_out;         /* number of output columns computed (conv domain) */
    int32_t batch;
    int32_t in_act;        /* TTSAMD_ACT_NONE | TTSAMD_ACT_LRELU, applied once per staged element */
    float in_slope;
    const float *in_mask;  /* [batch, t_in] multiplied into x before the activation, or NULL */
    int32_t mode;          /* TTSAMD_CONV_* */
    int32_t out_act;       /* v = act(acc + bias) */
    const float *res;      /* v += res[b,co,t]   (or NULL) */
    int64_t res_bstride, res_rstride;
    const float *accum;    /* v = accum[b,co,t] + v (or NULL) */
    int64_t accum_bstride, accum_rstride;
    const float *out_mask; /* [batch, t_out]: v *= mask (or NULL) */
    float out_div;         /* v = v / out_div when != 0 (true division: z_sum / num_kernels) */
    int32_t shuffle_u, shuffle_pad, shuffle_t_out; /* TTSAMD_CONV_SHUFFLE only */
    float *y2;             /* TTSAMD_CONV_RES_SKIP: second output (skip accumulator) */
    int64_t y2_bstride, y2_rstride;
    int32_t split_row;
    const float *row_bias; /* [batch, c_out] per-(b,row) additive term (speaker conditioning), or NULL */
    const void *w_split;   /* from ttsamd_conv1d_pack_weights_split, or NULL.  Non-NULL selects the split-bf16 kernels:
                            * both fp32 operands are split three ways into bf16 and the six leading products are
                            * accumulated in fp32 on the bf16 matrix pipe (every product is more accurate than one
                            * fp32 rounding of it); NULL = fp32-input MFMA kernels reading w_packed. */
    const void *w_h2;      /* ABI v3: from ttsamd_conv1d_pack_weights_h2, or NULL.  Non-NULL (together with w_split) lets the
                            * LARGE-GRID launches of the tuned (kernel, dilation) pairs run the three-product arithmetic: both fp32
                            * operands split into two fp16 parts (hi + lo * 2^-11, 22 significand bits + sign), hi*hi in one fp32
                            * accumulator, the two cross products in a second one that is scaled once in the epilogue; weights
                            * carry one power-of-two exponent per packed row (in the image), activations one per staged tile
                            * (derived in the kernel), both undone exactly.  fp32-class like the six-product scheme (error against
                            * fp64 1-2e-7 of sum|w x| on adversarial operands) at half its matrix-pipe work.  Small-grid launches
                            * (single requests) and the generic kernel keep the six-product split-bf16 arithmetic of w_split. */
} ttsamd_conv1d_args;

/* Dispatch notes: c_out == 1, kernel 7, dilation 1, NORMAL mode without residual / accumulate / masks on the output
 * (HiFiGAN conv_post, hifigan_generator.py:262-264) runs as an HBM-streaming kernel in exact fp32 whatever w_split is;
 * everything else runs on the matrix pipe as described above. */
int ttsamd_conv1d(const ttsamd_conv1d_args *args /* host */, void *stream);

/* Number of floats of the packed image of a [c_out, c_in, kernel] weight. */
size_t ttsamd_conv1d_packed_floats(int c_out, int c_in, int kernel);
/* HOST-side repack (load time): w [c_out, c_in, kernel] row-major (host) -> MFMA fragment order
 * dst (host) = [ceil(c_out/32)][k-step groups][64 lanes][4]; zero padded. */
int ttsamd_conv1d_pack_weights(float *dst, const float *w, int c_out, int c_in, int kernel);
/* Split-bf16 weight image (see ttsamd_conv1d_args.w_split): bytes, and the HOST-side repack
 * w [c_out, c_in, kernel] fp32 -> [ceil(c_out/32)][ceil(c_in/16)][kernel][3 parts][64 lanes][8 bf16] (+ two zero
 * groups of prefetch slack); part q of a weight = round-to-nearest bf16 of what parts < q left of it. */
size_t ttsamd_conv1d_packed_split_bytes(int c_out, int c_in, int kernel);
int ttsamd_conv1d_pack_weights_split(void *dst, const float *w, int c_out, int c_in, int kernel);
/* Two-part fp16 weight image (see ttsamd_conv1d_args.w_h2): bytes, and the HOST-side repack
 * w [c_out, c_in, kernel] fp32 -> [ceil(c_out/32)][ceil(c_in/16)][kernel][2 parts][64 lanes][8 fp16] (+ two zero groups of
 * prefetch slack), then the row table: {int max_row_exp, 3 pad} and per packed row {float 2^e, float 2^-e}.  Row r is
 * multiplied by 2^e_r (its largest magnitude lands in [2^13, 2^14)) before the split: hi = fp16(w 2^e), lo = fp16((w 2^e - hi) 2^11). */
size_t ttsamd_conv1d_packed_h2_bytes(int c_out, int c_in, int kernel);
int ttsamd_conv1d_pack_weights_h2(void *dst, const float *w, int c_out, int c_in, int kernel);
/* 1 if ttsamd_conv1d takes this (kernel, dilation): any kernel <= 31 at any dilation <= 27.  ttsamd_conv1d_tuned: 1 for the pairs
 * with tuned template instantiations (k in {1,2,5} at d = 1; k in {3,7,11} at d in {1,3,5}; k = 3, d = 9); every other pair — and a
 * tuned pair in a mode it has no instantiation for — runs on a generic split-bf16 kernel (conv_generic.hip: one wave per 32x32
 * tile, correct for any reference-legal config, not fast): NORMAL, GATE and SHUFFLE modes, split image (w_split) required. */
int ttsamd_conv1d_supported(int kernel, int dilation);
int ttsamd_conv1d_tuned(int kernel, int dilation);
/* Launches that would put fewer than ~100 blocks on the chip (single-sentence requests): 0 = the large-grid tiles
 * everywhere, 1 = 64-column tiles with one 32x32 tile per wave (same summation order: bitwise the large-grid result),
 * 2 = those tiles plus, for c_in >= 128, wave groups that split the block's K loop and are reduced in a fixed order
 * (deterministic; fp32 reassociation relative to modes 0 / 1), 3 = as 2, with the small-grid kernels of
 * conv_kernel_x3s.h for kernel sizes <= 5 at dilation 1: a wave per 32x32 tile and K slice, weights requested a whole
 * K iteration ahead, straight-line request streams; 4 (default) = as 3, with the one-shot kernels of conv_kernel_x3o.h
 * first where the reduction fits 16 waves (dilation 1, k in {1,3,5,7}, up to 640 32x32 tiles): no K loop, the epilogue
 * spread over four waves.  Returns the previous mode. */
int ttsamd_conv1d_set_small_grid(int mode);

/* One ResBlock1 iteration of the HiFiGAN MRF as a single launch — replaces the body of the loop in
 * TTS/vocoder/models/hifigan_generator.py:90-98 (ResBlock1.forward):
 *     xt = F.leaky_relu(x, slope); xt = c1(xt)   [kernel k, dilation d, same padding]
 *     xt = F.leaky_relu(xt, slope); xt = c2(xt)  [kernel k, dilation 1]
 *     x  = xt + x
 * plus, for the block's last iteration, the MRF accumulate / average of hifigan_generator.py:255-261
 * (y = accum + y, then y = y / out_div when out_div != 0).  x, y, accum: [batch, c, t] fp32 contiguous (y may not
 * alias x); mask [batch, t] (or NULL) multiplies the input of BOTH convs (ragged-exact batching, see ttsamd_conv1d_args
 * in_mask).  Weights are the split-bf16 images of ttsamd_conv1d_pack_weights_split ([c, c, k] each); the intermediate
 * tensor stays in LDS (5 HBM tensor passes -> 2) and the arithmetic is product-for-product that of two ttsamd_conv1d
 * launches with w_split set: results are bitwise identical to the unfused pair.
 * Limits: c in {8, 16, 32, 64, 128} (and 256 with the two-part fp16 images), kernel in {3, 7, 11}, dilation in {1, 3, 5}
 * (ttsamd_resblock_pair_supported / ttsamd_resblock_pair_h2_supported).
 * c = 8 / 16 (HiFiGAN-v2's late stages) run on the 32-channel tile: pass the split images of the weights ZERO-PADDED to
 * [32, 32, k] (tensors and biases keep their real channel count; equal to the unfused pair up to the sign of zeros). */
typedef struct ttsamd_resblock_args {
    const float *x;
    float *y;
    const float *accum;     /* or NULL */
    const float *mask;      /* [batch, t] or NULL */
    const void *w1_split;
    const float *bias1;     /* [c] or NULL */
    const void *w2_split;
    const float *bias2;
    int32_t c, t, batch;
    int32_t kernel, dilation;
    float slope;            /* leaky-ReLU slope of both activations */
    float out_div;
    int32_t variant;        /* 0 = default tile; other values select alternative tiles (measurement only) */
    int64_t w1_bytes, w2_bytes; /* ABI v2: size of each split image; must equal ttsamd_resblock_weight_bytes(c, kernel) — the
                                   kernel walks whole 32-channel tiles (c = 8 / 16: the image of the [32, 32, k] padded weight) */
    const void *w1_h2, *w2_h2;  /* ABI v3: the two-part fp16 images of the same weights (ttsamd_conv1d_pack_weights_h2 of the [c, c, k]
                                   — c < 32: zero-padded [32, 32, k] — weight), or NULL.  Both non-NULL: launches on the default tiles
                                   run the three-product arithmetic of ttsamd_conv1d_args.w_h2 (one power-of-two exponent per block
                                   for the x tile and one for the intermediate tile, derived in the kernel); results then agree with
                                   the unfused pair to fp32 rounding level instead of bitwise.  The narrow small-grid tiles (single
                                   sentences) and ttsamd_resblock_group keep the six-product images. */
    int64_t w1_h2_bytes, w2_h2_bytes; /* must equal ttsamd_resblock_weight_h2_bytes(c, kernel) when the images are given */
} ttsamd_resblock_args;
/* bytes of the split-bf16 weight image ttsamd_resblock_pair reads for a c-channel pair (= ttsamd_conv1d_packed_split_bytes of
 * the [max(c,32), max(c,32), kernel] weight) */
size_t ttsamd_resblock_weight_bytes(int c, int kernel);
size_t ttsamd_resblock_weight_h2_bytes(int c, int kernel);   /* = ttsamd_conv1d_packed_h2_bytes of the [max(c,32), max(c,32), kernel] weight */
int ttsamd_resblock_pair(const ttsamd_resblock_args *args /* host */, void *stream);
int ttsamd_resblock_pair_supported(int c, int kernel, int dilation);
/* the same question for a call that carries the two-part fp16 images (w1_h2 / w2_h2): additionally c = 256 (ABI v4, round 6) */
int ttsamd_resblock_pair_h2_supported(int c, int kernel, int dilation);
/* The three branches of one MRF stage at one dilation — kernel sizes 3, 7, 11 in slots 0, 1, 2 of args3 (a slot with x == NULL is
 * absent), same c / t / batch / dilation — as ONE launch (hifigan_generator.py:255-261: the resblocks of a stage are independent).
 * For the small grids of a single sentence only (ttsamd_resblock_group_supported: c in {8,16,32,64} and t * batch within the
 * narrow-tile range of ttsamd_resblock_pair): there a stage's nine 10-20 us launches on three branch streams spend as long in
 * cross-stream joins as in kernels.  Each slot computes exactly what ttsamd_resblock_pair computes for it (bitwise). */
int ttsamd_resblock_group(const ttsamd_resblock_args *args3 /* host, [3] */, void *stream);
int ttsamd_resblock_group_supported(int c, int t, int batch);
/* y[i] = ((a[i] + b[i]) [+ c[i]]) / div — z_sum / num_kernels over branch outputs written separately (c may be NULL; n % 4 == 0,
 * 16-byte aligned). */
int ttsamd_sum_div(float *y, const float *a, const float *b, const float *c, float div, int64_t n, void *stream);

/* ------------------------------------------------------------------------------------------
 * Model-level handle of the vocoder — replaces, as ONE object a non-Python host can drive:
 *   TTS/vocoder/models/hifigan_generator.py:162-234  HifiganGenerator.__init__ (layer list from the config)
 *   :284-301  remove_weight_norm / load_checkpoint   (weight-norm folded at ttsamd_hifigan_finalize)
 *   :236-265  forward,  :267-282  inference          (replicate padding of `inference_padding` frames, no crop)
 *   TTS/vocoder/models/gan.py:58-66  GAN.inference   (the wrapper around it)
 * create(config) -> load(name, tensor) for every state_dict entry in the REFERENCE's key layout ("conv_pre.weight" or
 * "conv_pre.parametrizations.weight.original0/1" or ".weight_g/.weight_v", "ups.0...", "resblocks.3.convs1.2...", "conv_post...",
 * ".bias") -> finalize (fold weight norm as torch does — over every dim but 0, i.e. per in_channel for ConvTranspose1d —,
 * polyphase form of the transposed convs, the three fragment images of every conv, upload) -> forward(mel) any number of times
 * -> destroy.  forward launches exactly the kernel sequence of the Python host (tts_amd/hifigan.py: same ttsamd_conv1d /
 * ttsamd_resblock_pair / ttsamd_resblock_group calls, same tile choices — bitwise the same waveform, tests/test_hifigan_gpu.py),
 * on ONE stream, into a workspace the handle owns; with use_graph != 0 the sequence for a given (mel, lengths, wav, batch, frames,
 * stream) is captured into a hipGraph on its first call and replayed afterwards (a single sentence is ~100 launches of a few
 * microseconds: replay removes the host from the loop).  Speaker conditioning (cond_layer / XTTS conds[i]) is not part of the
 * handle: those models run through the kernel-level ABI.
 * Caller owns mel [batch, in_channels, frames] / lengths [batch] int64 or NULL (ragged-exact batching: row b is computed as if
 * it were alone with lengths[b] frames; needs k - stride even in every upsample layer) / wav
 * [batch, out_channels, ttsamd_hifigan_output_samples(frames)], all on the device.  One caller at a time per handle. */
#define TTSAMD_HIFIGAN_MAX_KERNELS 8
#define TTSAMD_HIFIGAN_MAX_DILATIONS 8
#define TTSAMD_HIFIGAN_MAX_UPSAMPLES 12
typedef struct ttsamd_hifigan_config {
    int32_t in_channels, out_channels;       /* HifiganGenerator(in_channels, out_channels, ...), hifigan_generator.py:163-178 */
    int32_t resblock_type;                   /* 1 = ResBlock1 (:18-105), 2 = ResBlock2 (:108-159) */
    int32_t num_kernels;
    int32_t resblock_kernel_sizes[TTSAMD_HIFIGAN_MAX_KERNELS];
    int32_t num_dilations[TTSAMD_HIFIGAN_MAX_KERNELS];
    int32_t resblock_dilation_sizes[TTSAMD_HIFIGAN_MAX_KERNELS][TTSAMD_HIFIGAN_MAX_DILATIONS];
    int32_t num_upsamples;
    int32_t upsample_factors[TTSAMD_HIFIGAN_MAX_UPSAMPLES];
    int32_t upsample_kernel_sizes[TTSAMD_HIFIGAN_MAX_UPSAMPLES];
    int32_t upsample_initial_channel;
    int32_t inference_padding;               /* frames replicated either side by `inference` (:267-282); 0 = plain forward */
    int32_t precision;                       /* conv arithmetic: 0 = three fp16 products on large grids (default of the Python host),
                                              * 1 = six bf16 products, 2 = fp32-input MFMA */
} ttsamd_hifigan_config;
int ttsamd_hifigan_create(const ttsamd_hifigan_config *config /* host */, void **handle_out);
/* one state_dict entry: `data` (host, fp32, row-major) of `shape[ndim]` under the reference's key `name` */
int ttsamd_hifigan_load(void *handle, const char *name, const float *data, const int64_t *shape, int ndim);
int ttsamd_hifigan_finalize(void *handle);
/* output samples per item for `frames` input frames: (frames + 2 * inference_padding) * hop when every k - stride is even */
int64_t ttsamd_hifigan_output_samples(void *handle, int frames);
int ttsamd_hifigan_forward(void *handle, const float *mel, int batch, int frames, const int64_t *lengths, float *wav, int use_graph,
                           void *stream);
/* The same with `in_mask` [batch, frames] (device, or NULL) multiplied into the input inside conv_pre's load: the waveform decoder
 * inside VITS is fed `z * y_mask` (TTS/tts/models/vits.py:1161).  `lengths` and `in_mask` are alternatives. */
int ttsamd_hifigan_forward_ex(void *handle, const float *mel, int batch, int frames, const int64_t *lengths, const float *in_mask,
                              float *wav, int use_graph, void *stream);
/* Options of a vocoder handle.  CONCURRENT_BRANCHES (default 1): the MRF resblocks of a stage run on the handle's own branch streams
 * (one per resblock kernel size), joined by events in the reference's accumulation order — right for a lone request, whose branch
 * launches are too small to fill the chip; a host that keeps several requests in flight (one handle per request lane) sets 0. */
#define TTSAMD_HIFIGAN_OPT_CONCURRENT_BRANCHES 1
int ttsamd_hifigan_set_option(void *handle, int option, int value);
int ttsamd_hifigan_destroy(void *handle);

/* ------------------------------------------------------------------------------------------
 * Model-level handle of VITS ("mi355_vits_{create,load,infer,destroy}") — replaces, as ONE object a non-Python
 * host can drive:
 *   TTS/tts/models/vits.py:653-724    Vits.__init__ layer wiring (text encoder, duration predictor, flow, waveform decoder)
 *   :1698-1725                        load_checkpoint (weight norm stays parametrised in the checkpoint; folded at finalize)
 *   :1088-1173                        Vits.inference: text encoder -> duration predictor (reverse) -> ceil / cumsum ->
 *                                     generate_path -> prior expansion + noise -> flow (reverse) -> waveform decoder
 *   TTS/tts/layers/vits/networks.py:29-100,103-232; stochastic_duration_predictor.py:150-294; glow_tts/duration_predictor.py:7-69;
 *   glow_tts/transformer.py:10-432; generic/wavenet.py:16-123 (the layers behind it)
 * Envelope: the single-speaker, single-language model of BASELINE.json (LJSpeech VITS): no speaker / language conditioning, no
 * encoder_sample_rate interpolation, mean-only coupling — the Python host (tts_amd/vits.py) covers the rest through the kernel-level
 * ABI.  A request is two calls, because the output extent is data dependent and the caller owns every buffer:
 *   encode: tokens -> durations; returns when y_lengths are on the host (the request's ONE host wait, a pinned mirror the durations
 *           kernel writes) -> the caller sizes its outputs and draws noise_z at the reference's shape [batch, hidden, t_dec]
 *           (randn_like(m_p), vits.py:1155)
 *   decode: prior expansion, flows, waveform decoder into the caller's buffers.
 * Both issue exactly the kernel-level ABI calls of the Python host (same kernels, same tiles: bitwise the same outputs for the same
 * folded weights, tests/test_vits_gpu.py).  One caller, one request at a time per handle; errors are return codes. */
typedef struct ttsamd_vits_config {
    int32_t num_chars;                         /* rows of text_encoder.emb (VitsArgs.num_chars, vits.py:544-600) */
    int32_t hidden_channels;                   /* 192 */
    int32_t hidden_channels_ffn_text_encoder;  /* 768 */
    int32_t num_heads_text_encoder;            /* 2 */
    int32_t num_layers_text_encoder;           /* 6 */
    int32_t kernel_size_text_encoder;          /* 3 */
    int32_t kernel_size_flow;                  /* 5 */
    int32_t dilation_rate_flow;                /* 1 */
    int32_t num_layers_flow;                   /* 4: WaveNet layers per coupling block */
    int32_t num_flows;                         /* 4: ResidualCouplingBlocks(num_flows=4), networks.py:190; must be even */
    int32_t use_sdp;                           /* 1: StochasticDurationPredictor(hidden, 192, 3, p, 4) (vits.py:684-692); 0: DurationPredictor(hidden, 256, 3, p) */
    float inference_noise_scale;               /* 0.667 */
    float inference_noise_scale_dp;            /* 1.0 */
    float length_scale;                        /* 1.0 */
    ttsamd_hifigan_config decoder;             /* waveform_decoder (vits.py:704-718): in_channels = hidden_channels, inference_padding 0 */
} ttsamd_vits_config;
/* every output is optional (NULL = not wanted) except wav; all device pointers, fp32 unless noted, contiguous */
typedef struct ttsamd_vits_outputs {
    float *wav;          /* [batch, 1, t_dec * hop]   "model_outputs" */
    float *alignments;   /* [batch, t_text, t_dec]    "alignments" */
    float *durations;    /* [batch, t_text]           "durations" (w_ceil) */
    float *z, *z_p, *m_p, *logs_p;   /* [batch, hidden, t_dec] */
    float *y_mask;       /* [batch, t_dec] */
    int64_t *y_lengths;  /* [batch] int64 */
    float *logw;         /* [batch, t_text]: the duration predictor's output (NULL when it did not run) */
    float *x_hidden;     /* [batch, hidden, t_text]: the text encoder's hidden output */
    int32_t t_text_out;  /* 0, or the token count the token-indexed outputs (alignments, durations, logw, x_hidden) are written at when
                            the request ran on a PADDED token axis (a host that pads ids to a length bucket so that requests share a
                            captured front end: pad ids masked out by x_lengths own no frames): [batch, t_text_out, t_dec] etc.
                            Only with a replayed tail (decode use_graph on a single request), whose outputs are copies anyway. */
} ttsamd_vits_outputs;
int ttsamd_vits_create(const ttsamd_vits_config *config /* host */, void **handle_out);
/* one state_dict entry under its reference key ("text_encoder.emb.weight", "duration_predictor.flows.1.convs.norms_1.0.gamma",
 * "flow.flows.0.enc.in_layers.0.parametrizations.weight.original0", "waveform_decoder.ups.0...", ...); "disc.*" keys are ignored */
int ttsamd_vits_load(void *handle, const char *name, const float *data /* host */, const int64_t *shape /* host */, int ndim);
int ttsamd_vits_finalize(void *handle);
/* x int64 [batch, t_text], x_lengths int64 [batch] (device).  noise_dp [batch, 2, t_text] (device): the SDP's randn draw
 * (stochastic_duration_predictor.py:287), required when use_sdp and the predictor runs.  durations_in [batch, t_text] (device) or
 * NULL: injected durations (vits.py:1141-1143); the predictor then runs only if run_duration_predictor != 0.  On return
 * y_lengths_host[batch] (host, may be NULL) and *t_dec_out hold every item's frame count and their maximum. */
int ttsamd_vits_encode(void *handle, const int64_t *x, const int64_t *x_lengths, int batch, int t_text, const float *noise_dp,
                       const float *durations_in, int run_duration_predictor, int64_t *y_lengths_host, int32_t *t_dec_out,
                       int use_graph, void *stream);
/* second half of the request encode started.  noise_z [batch, hidden, t_dec] (device), t_dec as returned by encode.  use_graph != 0:
 * a single request (batch 1) replays its tail — prior expansion, flows, waveform decoder, ~110 launches — as one hipGraph per 32-frame
 * length bucket over the handle's static buffers, run ragged-exact inside the padded tensors, and copies the outputs out cut to the
 * true extent (one launch); batches run eagerly. */
int ttsamd_vits_decode(void *handle, const float *noise_z, const ttsamd_vits_outputs *out /* host */, int use_graph, void *stream);
int64_t ttsamd_vits_hop_length(void *handle);
int ttsamd_vits_set_option(void *handle, int option, int value);   /* forwarded to the waveform decoder: TTSAMD_HIFIGAN_OPT_* */
int ttsamd_vits_destroy(void *handle);

/* ------------------------------------------------------------------------------------------
 * Model-level handle of Glow-TTS ("mi355_glowtts_{...}") — replaces, as ONE object:
 *   TTS/tts/models/glow_tts.py:59-105   GlowTTS.__init__ wiring (encoder, decoder), :519-530 store_inverse / load_checkpoint
 *   :341-374, :137-148                  GlowTTS.inference / compute_outputs
 *   TTS/tts/layers/glow_tts/encoder.py:15-179 (rel_pos_transformer encoder with prenet and DurationPredictor),
 *   glow_tts/decoder.py:8-141, glow.py:11-233, generic/normalization.py:5-123 (squeeze, 12 x [ActNorm, InvConvNear, CouplingBlock], unsqueeze)
 * Envelope: the single-speaker model of BASELINE configs[0] (GlowTTSConfig defaults, glow_tts_config.py:101-152): no speaker
 * conditioning, sigmoid_scale False, num_splits 4.  Same two-call request shape as the VITS handle: encode returns the output
 * extent; decode writes the mel (channels first [batch, out_channels, t_mel], t_mel = t_dec / num_squeeze * num_squeeze — the
 * reference's [B, T, C] view is a transpose) and the other compute_outputs tensors into the caller's buffers. */
typedef struct ttsamd_glowtts_config {
    int32_t num_chars;
    int32_t hidden_channels_enc;        /* 192 */
    int32_t hidden_channels_dec;        /* 192 */
    int32_t hidden_channels_dp;         /* 256 */
    int32_t out_channels;               /* 80 */
    int32_t encoder_kernel_size, encoder_num_layers, encoder_num_heads, encoder_hidden_channels_ffn;   /* encoder_params: 3, 6, 2, 768 */
    int32_t encoder_rel_attn_window_size;   /* 0 = None (the config default): plain attention */
    int32_t encoder_layer_norm_type;    /* 1 (eps 1e-4, the default) or 2 (eps 1e-5) */
    int32_t use_encoder_prenet;         /* 1 */
    int32_t mean_only;                  /* 1 */
    int32_t num_flow_blocks_dec;        /* 12 */
    int32_t kernel_size_dec, dilation_rate, num_block_layers;   /* 5, 1, 4 */
    int32_t num_splits, num_squeeze;    /* 4, 2 */
    float inference_noise_scale;        /* 0.0 (glow_tts_config.py:151) */
    float length_scale;                 /* 1.0 */
    int32_t precision;                  /* 0 h2, 1 x3, 2 f32 (as ttsamd_hifigan_config.precision) */
} ttsamd_glowtts_config;
typedef struct ttsamd_glowtts_outputs {
    float *mel;            /* [batch, out_channels, t_mel]  "model_outputs" (transposed) */
    float *y_mean;         /* [batch, out_channels, t_dec] */
    float *y_log_scale;    /* [batch, out_channels, t_dec] */
    float *alignments;     /* [batch, t_text, t_dec] */
    float *durations_log;  /* [batch, t_text] */
    float *total_durations_log;   /* [batch, t_text] */
    float *durations;      /* [batch, t_text]  (w_ceil) */
    int64_t *y_lengths;    /* [batch] */
} ttsamd_glowtts_outputs;
int ttsamd_glowtts_create(const ttsamd_glowtts_config *config /* host */, void **handle_out);
int ttsamd_glowtts_load(void *handle, const char *name, const float *data /* host */, const int64_t *shape /* host */, int ndim);
int ttsamd_glowtts_finalize(void *handle);
/* x int64 [batch, t_text], x_lengths int64 [batch] (device); durations_in [batch, t_text] or NULL (a parity harness pins the integer
 * durations with it).  ragged_exact != 0: padded tokens own no frames (the reference gives each PADDED token one frame via
 * clamp_min, which only matters in batches). */
int ttsamd_glowtts_encode(void *handle, const int64_t *x, const int64_t *x_lengths, int batch, int t_text, const float *durations_in,
                          int ragged_exact, int64_t *y_lengths_host, int32_t *t_dec_out, int use_graph, void *stream);
/* noise [batch, out_channels, t_dec] (device) or NULL when inference_noise_scale == 0 */
int ttsamd_glowtts_decode(void *handle, const float *noise, const ttsamd_glowtts_outputs *out /* host */, void *stream);
int ttsamd_glowtts_destroy(void *handle);

/* ------------------------------------------------------------------------------------------
 * Channel LayerNorm on [B, C, T] (normalise over C for every (b, t)), with the fusions the text
 * encoder / duration predictors need.
 * replaces: TTS/tts/layers/generic/normalization.py:5-28 (LayerNorm, eps 1e-4) and :31-53
 *   (LayerNorm2, eps 1e-5); the depthwise conv + norm + GELU chain of DilatedDepthSeparableConv
 *   (TTS/tts/layers/vits/stochastic_duration_predictor.py:46-63); the residual adds around the
 *   norms in TTS/tts/layers/glow_tts/transformer.py:419-431.
 *   u[c,t]  = dw ? dw_bias[c] + sum_k dw_w[c,k] * (x*in_mask)[c, t + (k-(K-1)/2)*dil] : x[c,t]
 *   u      += pre_res[c,t]                                   (if pre_res)
 *   v       = (u - mean_c(u)) * rsqrt(var_c(u) + eps) * gamma[c] + beta[c]   (biased variance)
 *   v       = act(v)   (NONE | RELU | GELU(erf))
 *   v       = post_res[c,t] + v                              (if post_res)
 *   y[c,t]  = v * out_mask[t]                                (if out_mask)
 * Any channel count (up to 512 a thread keeps its channels in registers between the passes; beyond, the passes re-read x:
 * y must then not alias x). */
typedef struct ttsamd_norm_args {
    const float *x;
    int64_t x_bstride, x_rstride;
    int32_t c, t, batch;
    const float *gamma, *beta; /* [c] */
    float eps;
    const float *dw_w, *dw_bias; /* [c, dw_kernel], [c] or NULL */
    int32_t dw_kernel, dw_dilation;
    const float *in_mask; /* [batch, t], used by the depthwise prologue */
    const float *pre_res;
    int64_t pre_bstride, pre_rstride;
    int32_t act;
    const float *post_res;
    int64_t post_bstride, post_rstride;
    const float *out_mask; /* [batch, t] */
    float *y;
    int64_t y_bstride, y_rstride;
} ttsamd_norm_args;
int ttsamd_channel_norm(const ttsamd_norm_args *args /* host */, void *stream);

/* ------------------------------------------------------------------------------------------
 * Relative-position multi-head attention core (everything between conv_q/k/v and conv_o).
 * replaces: RelativePositionMultiHeadAttention.attention, TTS/tts/layers/glow_tts/transformer.py:118-163
 *   incl. _get_relative_embeddings / _relative_position_to_absolute_position /
 *   _absolute_position_to_relative_position (:196-241).
 *   q,k,v: [B, H*dk, T] (row stride t, batch stride qkv_bstride — they may live in one fused
 *   [B, 3*H*dk, T] projection buffer); mask [B, T] (x_mask; attn_mask[i][j] = mask[i]*mask[j],
 *   masked scores = -1e4, transformer.py:147); emb_rel_k / emb_rel_v [2*window+1, dk] shared by
 *   all heads (heads_share=True) or NULL when rel_attn_window_size is None (window ignored).
 *   out [B, H*dk, T] contiguous.  QK^T and P.V run on the fp32-input MFMA (exact fp32 products).
 * Limits: dk <= 128 (any value: channels are zero-padded to the next multiple of 32 inside the kernel; multilingual
 * VITS has dk = 98).  No limit on T: up to 1024 the [32][T] score strip of a query block lives in LDS; beyond that the
 * scores are recomputed tile by tile around an online softmax (same results up to the summation order of the softmax
 * denominator). */
int ttsamd_rel_attention(float *out, const float *q, const float *k, const float *v, int64_t qkv_bstride,
                         const float *mask, const float *emb_rel_k, const float *emb_rel_v, int window,
                         int batch, int heads, int dk, int t, void *stream);

/* ------------------------------------------------------------------------------------------
 * Text-side small kernels (HBM/latency-bound; one coalesced pass each)
 * ---------------------------------------------------------------------------------------- */
/* y[b,c,t] = emb[tokens[b,t], c] * scale * mask[b,t]     (TextEncoder.forward, networks.py:87-96;
 * glow_tts/encoder.py:156-160).  tokens int64 [B,T], emb [V,C], mask [B,T] or NULL. */
int ttsamd_embed(float *y, const int64_t *tokens, const float *emb, const float *mask, float scale,
                 int batch, int c, int t, int vocab, void *stream);
/* Multilingual text encoder input (networks.py:87-93): y [B, c + c_extra, T]; rows 0..c-1 as ttsamd_embed, rows
 * c.. = extra[b, :] (the language embedding `emb_l(lid)`, vits.py:1119-1124) broadcast over time, unscaled, * mask. */
int ttsamd_embed_cat(float *y, const int64_t *tokens, const float *emb, const float *mask, float scale,
                     const float *extra, int c_extra, int batch, int c, int t, int vocab, void *stream);

/* mask[b,t] = t < lengths[b] ? 1 : 0   (sequence_mask, TTS/tts/utils/helpers.py:43-57). lengths int64 [B]. */
int ttsamd_sequence_mask(float *mask, const int64_t *lengths, int batch, int t, void *stream);

/* ConvFlow head (stochastic_duration_predictor.py:121-122 + DDSConv's `x = x + g`, :55-56):
 *   h[b,c,t] = w[c] * z[b, z_ch, t] + bias[c] + g[b,c,t]      (pre = Conv1d(1, C, 1)) */
int ttsamd_convflow_pre(float *h, const float *z, int z_ch, const float *w, const float *bias, const float *g,
                        int batch, int c, int t, void *stream);

/* ConvFlow tail, reverse direction (stochastic_duration_predictor.py:126-147 +
 * TTS/tts/layers/vits/transforms.py:12-202, inverse=True, tails="linear", tail_bound):
 *   params h [B, 3*bins-1, T] (proj output, already masked), z_in [B,2,T];
 *   the flow sees zf = flip(z_in, channel) (stochastic_duration_predictor.py:289): x0 = z_in[:,1], x1 = z_in[:,0];
 *   z_out[:,0] = x0 * mask,  z_out[:,1] = rq_spline_inverse(x1; h / sqrt(filter) ...) * mask. */
int ttsamd_convflow_spline_reverse(float *z_out, const float *z_in, const float *h, const float *mask,
                                   int batch, int t, int num_bins, float filter_channels, float tail_bound,
                                   void *stream);

/* ElementwiseAffine reverse after a channel flip (stochastic_duration_predictor.py:82-83,289):
 *   z_out[b,c,t] = (z_in[b,1-c,t] - m[c]) * exp(-logs[c]) * mask[b,t],  c in {0,1}. */
int ttsamd_sdp_affine_reverse(float *z_out, const float *z_in, const float *m, const float *logs,
                              const float *mask, int batch, int t, void *stream);

/* Durations from log-durations.
 * VITS (vits.py:1140-1146):   w = exp(logw) * mask * length_scale;  w_ceil = ceil(w)
 * Glow (glow_tts.py:350-352): w = (exp(logw) - 1) * mask * length_scale;  w_ceil = max(ceil(w), 1)   [glow != 0]
 *   durations [B,T] float (= w_ceil), cum [B,T] int32 inclusive cumsum, y_lengths int64 [B] = max(sum, 1).
 *   glow == 2: as Glow but padded tokens get 0 frames (w_ceil *= mask) — the reference counts one frame per PADDED token
 *   into y_lengths when batching; 2 makes a padded batch reproduce the per-sentence results.
 * `durations_in` (may be NULL) overrides w_ceil (logw ignored). */
int ttsamd_durations(float *durations, int32_t *cum, int64_t *y_lengths, const float *logw,
                     const float *durations_in, const float *mask, float length_scale, int glow, int batch,
                     int t, void *stream);
/* The same with two extras (ABI v2):
 *   t_valid: columns t >= t_valid own no frames whatever the rule (a request padded to a text-length bucket: the reference's
 *     clamp_min(.,1) still gives every column of the CALLER's tensor a frame, masked or not — glow_tts.py:350-351 — and the
 *     bucket's own padding none); pass t for "all columns are the caller's".
 *   y_lengths_host (or NULL): host-mapped (pinned) mirror of y_lengths, written with a system-scope release store — the host
 *     sets it to -1 before the call and polls it instead of a stream synchronise + device-to-host copy. */
int ttsamd_durations_ex(float *durations, int32_t *cum, int64_t *y_lengths, int64_t *y_lengths_host, const float *logw,
                        const float *durations_in, const float *mask, float length_scale, int glow, int t_valid,
                        int batch, int t, void *stream);

/* generate_path (helpers.py:154-169): attn[b,x,y] = (cum[b,x-1] <= y < cum[b,x]) * x_mask[b,x] * (y < y_lengths[b]). */
int ttsamd_generate_path(float *attn, const int32_t *cum, const float *x_mask, const int64_t *y_lengths,
                         int batch, int t_x, int t_y, void *stream);

/* Prior expansion: the two attn matmuls + the noise draw as one gather
 * (vits.py:1152-1155; glow_tts.py:137-148,361):
 *   x = token owning frame y (from cum);  valid = x_mask[b,x] && y < y_lengths[b]
 *   m_p[b,c,y] = valid ? m[b,c,x] : 0;  logs_p likewise (logs may be NULL = zeros: Glow mean_only)
 *   z_p[b,c,y] = (m_p + noise[b,c,y] * exp(logs_p) * noise_scale) * (mask_out ? y_mask : 1)
 *   y_mask[b,y] = y < y_lengths[b].   m_p / logs_p outputs may be NULL; z_p2 (may be NULL) gets a second copy. */
int ttsamd_expand_prior(float *z_p, float *z_p2, float *m_p, float *logs_p, float *y_mask, const float *m,
                        const float *logs, int64_t stats_bstride, const float *noise, const int32_t *cum,
                        const float *x_mask, const int64_t *y_lengths, float noise_scale, int mask_out, int batch,
                        int c, int t_x, int t_y, void *stream);
/* noise_packed != 0: `noise` is a contiguous [batch, c, max_b y_lengths[b]] tensor (the draw at the reference's shape,
 * randn_like(m_p), sitting at the head of a larger fixed buffer); columns beyond that extent take zero noise. */
int ttsamd_expand_prior_ex(float *z_p, float *z_p2, float *m_p, float *logs_p, float *y_mask, const float *m,
                           const float *logs, int64_t stats_bstride, const float *noise, const int32_t *cum,
                           const float *x_mask, const int64_t *y_lengths, float noise_scale, int mask_out, int noise_packed,
                           int batch, int c, int t_x, int t_y, void *stream);
/* m / logs are read as m[b*stats_bstride + c*t_x + x] (they are usually the two halves of one [B,2C,T_x]
 * projection buffer). */

/* y[i] = x[i] * s  (noise * noise_scale, stochastic_duration_predictor.py:287). */
int ttsamd_scale(float *y, const float *x, float s, int64_t n, void *stream);

/* ------------------------------------------------------------------------------------------
 * Glow-TTS decoder glue (HBM-bound permutations / 4-wide mixing; one pass each)
 * ---------------------------------------------------------------------------------------- */
/* squeeze (TTS/tts/layers/glow_tts/decoder.py:8-28): x [B,C,T] -> y [B,C*n,T/n] (an odd tail frame is dropped),
 * y[b, s*C+c, t'] = x[b,c,t'*n+s] * mask[b, t'*n+n-1];  mask_out [B,T/n] = mask[:, n-1::n] (may be NULL). */
int ttsamd_glow_squeeze(float *y, float *mask_out, const float *x, const float *mask, int batch, int c, int t, int n,
                        void *stream);
/* unsqueeze (decoder.py:31-47): x [B,Cq,Tq] -> y [B,Cq/n,t_out], y[b,c,t'*n+s] = x[b, s*(Cq/n)+c, t'] * mask_q[b,t'];
 * columns >= Tq*n (the dropped odd frame) are written as zeros. */
int ttsamd_glow_unsqueeze(float *y, const float *x, const float *mask_q, int batch, int cq, int tq, int n, int t_out,
                          void *stream);
/* InvConvNear reverse (TTS/tts/layers/glow_tts/glow.py:107-137, stored 4x4 inverse `w_inv` row-major) followed by
 * ActNorm reverse (TTS/tts/layers/generic/normalization.py:98-101; bias/logs [C], or both NULL to skip), IN PLACE
 * on x [B,C,T]:  z = (w_inv . x_group) * mask;  x = (z - bias) * exp(-logs) * mask.   num_splits must be 4.
 * forward != 0 runs the forward flow instead (ActNorm then InvConvNear, glow_tts/decoder.py:126-131):
 *   x = (bias + exp(logs) * x) * mask;  x = (w . x_group) * mask   with `w_inv` holding the weight itself. */
int ttsamd_glow_invconv_actnorm(float *x, const float *w_inv, const float *bias, const float *logs, const float *mask,
                                int batch, int c, int t, int num_splits, int forward, void *stream);
/* o[r] = sum_t x[r,t]: token durations of a MAS alignment, `attn.sum(-1)` (glow_tts.py:147, vits.py:922). */
int ttsamd_row_sum(float *o, const float *x, int64_t rows, int t, void *stream);
/* o_attn_dur = log(1 + sum_y attn[b,x,y]) * x_mask (GlowTTS.compute_outputs, TTS/tts/models/glow_tts.py:147), with
 * the row sums taken from the cumulative durations.  o [B,T_x]. */
int ttsamd_attn_durations(float *o, const int32_t *cum, const float *x_mask, const int64_t *y_lengths, int batch,
                          int t_x, void *stream);

/* ------------------------------------------------------------------------------------------
 * Small streaming kernels (HBM-bound; coalesced, one pass)
 * ---------------------------------------------------------------------------------------- */
/* Request glue (round 4): a single sentence is a chain of ~250 dependent launches; the tensor.clone() / slice / transpose
 * copies a request hands out (one small launch each in a tensor library) travel as ONE launch.
 * Segment i copies a [d0, d1, d2] box: element (a, b, c) from src + (a*s0 + b*s1 + c*s2) to dst + (a*t0 + b*t1 + c*t2) (elements). */
#define TTSAMD_COPY_MAX_SEGS 12
typedef struct ttsamd_copy_seg {
    const void *src;
    void *dst;
    int32_t d0, d1, d2;
    int64_t s0, s1, s2;     /* source strides in elements */
    int64_t t0, t1, t2;     /* destination strides in elements */
    int32_t elem_bytes;     /* 4 or 8 */
} ttsamd_copy_seg;
int ttsamd_copy_strided(const ttsamd_copy_seg *segs /* host */, int n, void *stream);
/* All length masks of a ragged vocoder call in one launch (HifiganGenerator.forward with `lengths`: one [B, T_stage] mask per
 * up-sampling stage): len_eff[b] = lengths[b] / quantum * quantum + add (Glow's squeeze drops the frames that do not fill a
 * group, decoder.py:19-20; `add` = 2 * inference_padding), stage s: masks[off_s + b*t_stage[s] + t] = t < len_eff[b]*scales[s],
 * off_s = batch * sum_{r<s} t_stage[r].  len_out (or NULL) receives len_eff. */
#define TTSAMD_MASK_MAX_STAGES 8
int ttsamd_stage_masks(float *masks, int64_t *len_out, const int64_t *lengths, int batch, int quantum, int add,
                       const int32_t *scales /* host */, const int32_t *t_stage /* host */, int n_stages, void *stream);
/* y[b,c,:] = F.pad(x[b,c,:], (pad,pad), "replicate")  — HifiganGenerator.inference,
 * TTS/vocoder/models/hifigan_generator.py:281.  x [rows, t], y [rows, t + 2*pad]. */
int ttsamd_replicate_pad(float *y, const float *x, int64_t rows, int t, int pad, void *stream);
/* Ragged-batch form: item b of x [B,C,t] holds lengths[b] valid frames; y [B,C,t+2*pad] replicates item b's OWN first
 * and last valid frame (y[b,c,i] = x[b,c,clamp(i-pad, 0, lengths[b]-1)]), so that a padded batch reproduces what the
 * reference computes sentence by sentence (synthesizer.py:384). */
int ttsamd_replicate_pad_ragged(float *y, const float *x, const int64_t *lengths, int batch, int c, int t, int pad,
                                void *stream);
/* The same with item b's valid frame count given as lengths[b] + len_bias (the vocoder's padded lengths minus 2*pad). */
int ttsamd_replicate_pad_ragged_ex(float *y, const float *x, const int64_t *lengths, int64_t len_bias, int batch, int c,
                                   int t, int pad, void *stream);

/* The Synthesizer's Glow-TTS -> vocoder seam on the device: `vocoder_ap.normalize(tts_ap.denormalize(mel))`
 * (TTS/utils/synthesizer.py:412-416; AudioProcessor.normalize / denormalize, TTS/utils/audio/processor.py:259-336;
 * StandardScaler, TTS/tts/utils/helpers.py:14-39) — the reference does this in numpy after a D2H copy and copies the
 * result back.  x, y [B, C, T]; same op order as the numpy code, fp32.  `mean`/`scale` ([C] device pointers, both
 * or neither) select the mean-variance scaler path (`stats_path` models). */
typedef struct ttsamd_mel_norm {
    int32_t signal_norm, symmetric_norm, clip_norm;
    float max_norm, min_level_db, ref_level_db;
    const float *mean, *scale;
} ttsamd_mel_norm;
int ttsamd_mel_renorm(float *y, const float *x, const ttsamd_mel_norm *tts /* host */,
                      const ttsamd_mel_norm *voc /* host */, int batch, int c, int t, void *stream);

/* Posterior sample z = (mean + noise * exp(log_scale)) * mask (PosteriorEncoder.forward, TTS/tts/layers/vits/networks.py:286-287).
 * stats [B, 2C, T] = mean | log_scale (the `proj` output, already masked), noise / z [B, C, T], mask [B, T]. */
int ttsamd_sample_gaussian(float *z, const float *stats, const float *noise, const float *mask, int batch, int c, int t,
                           void *stream);

/* Speaker-conditioning helpers.
 * g = F.normalize(d_vectors) (Vits._set_cond_input, TTS/tts/models/vits.py:882): y[r,:] = x[r,:] / max(||x[r,:]||, eps). */
int ttsamd_l2_normalize(float *y, const float *x, int rows, int cols, float eps, void *stream);
/* y[r,t] = x[r,t] + row_bias[r] over rows = B*C  (DurationPredictor `x + cond(g)`, TTS/tts/layers/glow_tts/duration_predictor.py:58-59;
 * everywhere else the per-(b,channel) conditioning offset rides in a conv epilogue as ttsamd_conv1d_args.row_bias). */
int ttsamd_add_row_bias(float *y, const float *x, const float *row_bias, int64_t rows, int t, void *stream);

/* 1-D linear interpolation along time, exactly torch.nn.functional.interpolate(x, scale_factor=[s], mode="linear")
 * (align_corners=False, the given scale factor drives the coordinate map) — HifiDecoder.forward,
 * TTS/tts/layers/xtts/hifigan_decoder.py:688-700.  x [rows, t_in] -> y [rows, t_out], t_out = floor(t_in * s). */
int ttsamd_linear_interp(float *y, const float *x, int64_t rows, int t_in, int t_out, double scale_factor, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TTS_AMD_H */
