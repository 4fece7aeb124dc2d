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

const char *ttsamd_last_error(void);
/* ABI version (bumped on any signature change) and the gfx arch the library was compiled for. */
int ttsamd_abi_version(void);
const char *ttsamd_arch(void);

/* ------------------------------------------------------------------------------------------
 * Monotonic alignment search
 * replaces: TTS/tts/utils/monotonic_align/core.pyx:42-47  maximum_path_c(paths, values, t_xs,
 *           t_ys, max_neg_val=-1e9)   (and maximum_path_each, core.pyx:11-37)
 * ---------------------------------------------------------------------------------------- */

/* Exact mirror of `maximum_path_c`: `values` fp32 [b,t_x,t_y] is updated IN PLACE inside the DP
 * band, `paths` int32 [b,t_x,t_y] must be pre-zeroed by the caller (helpers.py:188) and receives
 * 1 on the chosen path.  t_xs/t_ys int32 [b] (device).  Bit-exact with the reference for every
 * valid input (1 <= t_xs[i] <= t_ys[i]); items with t_x<=0 or t_y<=0 are left untouched.
 * Limit: t_x <= 2048 (TTSAMD_ERR_UNSUPPORTED above). */
int ttsamd_maximum_path_c(int32_t *paths, float *values, const int32_t *t_xs, const int32_t *t_ys,
                          int b, int t_x, int t_y, float max_neg_val, void *stream);

/* Scratch bytes needed by ttsamd_maximum_path (direction bit-planes). */
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

#define TTSAMD_CONV_NORMAL 0
/* WN gate: packed row tile 2a = tanh channels [32a,32a+32), tile 2a+1 = sigmoid channels; writes
 * tanh(.)*sigmoid(.) to out channel 32a+i (wavenet.py:6-13). c_out = number of packed rows (2H). */
#define TTSAMD_CONV_GATE 1
/* ConvTranspose1d (kernel 2u, stride u, pad u/2) in polyphase form: packed row m = co*u + r holds
 * the 2-tap filter of phase r; column q is written to y[b, co, q*u + r - shuffle_pad]. */
#define TTSAMD_CONV_SHUFFLE 2
/* affine coupling, mean only (networks.py:164): y = (res - (conv+bias)*mask) * mask */
#define TTSAMD_CONV_COUPLE 3

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
} ttsamd_conv1d_args;

int ttsamd_conv1d(const ttsamd_conv1d_args *args /* host */, void *stream);

/* Number of floats of the packed image of a [c_out, c_in, kernel] weight. */
size_t ttsamd_conv1d_packed_floats(int c_out, int c_in, int kernel);
/* HOST-side repack (load time): w [c_out, c_in, kernel] row-major (host) -> MFMA fragment order
 * dst (host) = [ceil(c_out/32)][k-step groups][64 lanes][4]; zero padded. */
int ttsamd_conv1d_pack_weights(float *dst, const float *w, int c_out, int c_in, int kernel);
/* 1 if (kernel, dilation) has a tuned instantiation. */
int ttsamd_conv1d_supported(int kernel, int dilation);

/* ------------------------------------------------------------------------------------------
 * Small streaming kernels (HBM-bound; coalesced, one pass)
 * ---------------------------------------------------------------------------------------- */
/* y[b,c,:] = F.pad(x[b,c,:], (pad,pad), "replicate")  — HifiganGenerator.inference,
 * TTS/vocoder/models/hifigan_generator.py:281.  x [rows, t], y [rows, t + 2*pad]. */
int ttsamd_replicate_pad(float *y, const float *x, int64_t rows, int t, int pad, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TTS_AMD_H */
