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

#ifdef __cplusplus
}
#endif
#endif /* TTS_AMD_H */
