// TEST INFRASTRUCTURE ONLY — CPU stand-ins for the kernel-level entries of include/tts_amd.h that the model-level handles call.  A stub
// does no arithmetic: it READS every input element and WRITES every output element its arguments declare (the documented extents of
// the real kernel), so that with the malloc-backed "device" of hip_stub.cpp the address sanitizer checks the handles' workspace
// arithmetic, weight-image sizes and pointer bookkeeping on the CPU.  The few stubs whose outputs steer the host (sequence masks,
// durations) write plausible values.  The pack / policy functions are the real ones (csrc/pack_host.cpp).
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/tts_amd.h"

namespace ttsamd {
static char g_err[512];
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ttsamd
extern "C" const char *ttsamd_last_error(void) { return ttsamd::g_err; }

long g_stub_launches = 0;
static volatile double g_sink;

static void rd(const void *p, size_t bytes)
{
    if (!p) return;
    const unsigned char *c = static_cast<const unsigned char *>(p);
    unsigned s = 0;
    for (size_t i = 0; i < bytes; ++i) s += c[i];
    g_sink = g_sink + s;
}
static void rd2(const float *p, int64_t rows, int64_t rstride, int64_t cols)
{
    if (!p) return;
    for (int64_t r = 0; r < rows; ++r) rd(p + r * rstride, (size_t)cols * 4);
}
static void wr2(float *p, int64_t rows, int64_t rstride, int64_t cols, float v = 0.25f)
{
    if (!p) return;
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < cols; ++c) p[r * rstride + c] = v;
}
#define BAD(cond, msg)                  \
    do {                                \
        if (cond) {                     \
            ttsamd::set_error(msg);     \
            return TTSAMD_ERR_INVALID;  \
        }                               \
    } while (0)

extern "C" {

int ttsamd_conv1d(const ttsamd_conv1d_args *ap, void *)
{
    BAD(!ap || !ap->x || !ap->y || !ap->w_packed, "conv1d stub: NULL tensor");
    const ttsamd_conv1d_args &a = *ap;
    ++g_stub_launches;
    BAD(a.c_in <= 0 || a.c_out <= 0 || a.batch < 0 || a.t_in < 0 || a.t_out < 0, "conv1d stub: bad shape");
    rd(a.w_packed, ttsamd_conv1d_packed_floats(a.c_out, a.c_in, a.kernel) * 4);
    if (a.w_split) rd(a.w_split, ttsamd_conv1d_packed_split_bytes(a.c_out, a.c_in, a.kernel));
    if (a.w_h2) rd(a.w_h2, ttsamd_conv1d_packed_h2_bytes(a.c_out, a.c_in, a.kernel));
    rd(a.bias, (size_t)a.c_out * 4);
    int out_rows = a.c_out;
    if (a.mode == TTSAMD_CONV_GATE) out_rows = a.c_out / 2;
    if (a.mode == TTSAMD_CONV_RES_SKIP) out_rows = a.split_row;
    if (a.mode == TTSAMD_CONV_COUPLE_AFFINE || a.mode == TTSAMD_CONV_COUPLE_AFFINE_FWD || a.mode == TTSAMD_CONV_COUPLE_AFFINE_MIX) out_rows = a.split_row;
    for (int b = 0; b < a.batch; ++b) {
        rd2(a.x + b * a.x_bstride, a.c_in, a.x_rstride, a.t_in);
        if (a.in_mask) rd(a.in_mask + (int64_t)b * a.t_in, (size_t)a.t_in * 4);
        if (a.out_mask) rd(a.out_mask + (int64_t)b * a.t_out, (size_t)a.t_out * 4);
        if (a.row_bias) rd(a.row_bias + (int64_t)b * a.c_out, (size_t)a.c_out * 4);
        if (a.res) rd2(a.res + b * a.res_bstride, out_rows, a.res_rstride, a.t_out);
        if (a.mode == TTSAMD_CONV_SHUFFLE) {
            BAD(a.shuffle_u <= 0, "conv1d stub: SHUFFLE needs shuffle_u");
            const int rows = a.c_out / a.shuffle_u;
            for (int r = 0; r < rows; ++r)
                for (int64_t q = 0; q < a.t_out; ++q)
                    for (int ph = 0; ph < a.shuffle_u; ++ph) {
                        const int64_t col = q * a.shuffle_u + ph - a.shuffle_pad;
                        if (col >= 0 && col < a.shuffle_t_out) a.y[b * a.y_bstride + r * a.y_rstride + col] = 0.5f;
                    }
        } else if (a.mode == TTSAMD_CONV_RES_SKIP) {
            wr2(a.y + b * a.y_bstride, a.split_row, a.y_rstride, a.t_out);
            BAD(!a.y2, "conv1d stub: RES_SKIP needs y2");
            if (a.accum) rd2(a.accum + b * a.accum_bstride, a.c_out - a.split_row, a.accum_rstride, a.t_out);
            wr2(a.y2 + b * a.y2_bstride, a.c_out - a.split_row, a.y2_rstride, a.t_out);
        } else if (a.mode == TTSAMD_CONV_COUPLE_AFFINE_MIX) {
            BAD(!a.y2, "conv1d stub: MIX needs the parameter block in y2");
            rd(a.y2, (size_t)(16 + 4 * a.split_row) * 4);
            float *whole = a.y + b * a.y_bstride - (int64_t)a.split_row * a.y_rstride;      // in place on both halves
            rd2(whole, 2 * a.split_row, a.y_rstride, a.t_out);
            wr2(whole, 2 * a.split_row, a.y_rstride, a.t_out);
        } else {
            if (a.accum) rd2(a.accum + b * a.accum_bstride, out_rows, a.accum_rstride, a.t_out);
            wr2(a.y + b * a.y_bstride, out_rows, a.y_rstride, a.t_out);
        }
    }
    return TTSAMD_OK;
}

static int pair_touch(const ttsamd_resblock_args &a)
{
    BAD(!a.x || !a.y || !a.w1_split || !a.w2_split || a.x == a.y, "resblock_pair stub: bad tensors");
    BAD(a.w1_bytes != (int64_t)ttsamd_resblock_weight_bytes(a.c, a.kernel) || a.w2_bytes != a.w1_bytes, "resblock_pair stub: weight image size");
    rd(a.w1_split, (size_t)a.w1_bytes);
    rd(a.w2_split, (size_t)a.w2_bytes);
    if (a.w1_h2 || a.w2_h2) {
        BAD(!a.w1_h2 || !a.w2_h2 || a.w1_h2_bytes != (int64_t)ttsamd_resblock_weight_h2_bytes(a.c, a.kernel) || a.w2_h2_bytes != a.w1_h2_bytes,
            "resblock_pair stub: h2 image size");
        rd(a.w1_h2, (size_t)a.w1_h2_bytes);
        rd(a.w2_h2, (size_t)a.w2_h2_bytes);
    }
    rd(a.bias1, (size_t)a.c * 4);
    rd(a.bias2, (size_t)a.c * 4);
    const size_t n = (size_t)a.batch * a.c * a.t;
    rd(a.x, n * 4);
    rd(a.accum, n * 4);
    rd(a.mask, (size_t)a.batch * a.t * 4);
    wr2(a.y, 1, 0, (int64_t)n);
    return TTSAMD_OK;
}
int ttsamd_resblock_pair(const ttsamd_resblock_args *a, void *)
{
    BAD(!a, "resblock_pair stub: NULL args");
    ++g_stub_launches;
    return pair_touch(*a);
}
int ttsamd_resblock_group_supported(int, int, int) { return 0; }     // the handles then take the one-pair-per-launch path
int ttsamd_resblock_group(const ttsamd_resblock_args *a3, void *)
{
    ++g_stub_launches;
    for (int i = 0; i < 3; ++i)
        if (a3[i].x) {
            const int rc = pair_touch(a3[i]);
            if (rc) return rc;
        }
    return TTSAMD_OK;
}
int ttsamd_sum_div(float *y, const float *a, const float *b, const float *c, float, int64_t n, void *)
{
    ++g_stub_launches;
    rd(a, (size_t)n * 4), rd(b, (size_t)n * 4), rd(c, (size_t)n * 4);
    wr2(y, 1, 0, n);
    return TTSAMD_OK;
}
int ttsamd_stage_masks(float *masks, int64_t *len_out, const int64_t *lengths, int batch, int quantum, int add, const int32_t *scales, const int32_t *t_stage,
                       int n_stages, void *)
{
    ++g_stub_launches;
    BAD(n_stages < 1 || n_stages > TTSAMD_MASK_MAX_STAGES || quantum < 1, "stage_masks stub: bad arguments");
    size_t off = 0;
    for (int s = 0; s < n_stages; ++s) {
        for (int b = 0; b < batch; ++b) {
            const int64_t le = lengths[b] / quantum * quantum + add;
            for (int t = 0; t < t_stage[s]; ++t) masks[off + (size_t)b * t_stage[s] + t] = t < le * scales[s] ? 1.f : 0.f;
        }
        off += (size_t)batch * t_stage[s];
    }
    if (len_out)
        for (int b = 0; b < batch; ++b) len_out[b] = lengths[b] / quantum * quantum + add;
    return TTSAMD_OK;
}
int ttsamd_replicate_pad(float *y, const float *x, int64_t rows, int t, int pad, void *)
{
    ++g_stub_launches;
    rd(x, (size_t)rows * t * 4);
    wr2(y, rows, t + 2 * pad, t + 2 * pad);
    return TTSAMD_OK;
}
int ttsamd_replicate_pad_ragged_ex(float *y, const float *x, const int64_t *lengths, int64_t, int batch, int c, int t, int pad, void *)
{
    ++g_stub_launches;
    rd(lengths, (size_t)batch * 8);
    rd(x, (size_t)batch * c * t * 4);
    wr2(y, (int64_t)batch * c, t + 2 * pad, t + 2 * pad);
    return TTSAMD_OK;
}
int ttsamd_channel_norm(const ttsamd_norm_args *ap, void *)
{
    BAD(!ap || !ap->x || !ap->y || !ap->gamma || !ap->beta, "channel_norm stub: NULL tensor");
    const ttsamd_norm_args &a = *ap;
    ++g_stub_launches;
    rd(a.gamma, (size_t)a.c * 4), rd(a.beta, (size_t)a.c * 4);
    if (a.dw_w) rd(a.dw_w, (size_t)a.c * a.dw_kernel * 4), rd(a.dw_bias, (size_t)a.c * 4);
    for (int b = 0; b < a.batch; ++b) {
        rd2(a.x + b * a.x_bstride, a.c, a.x_rstride, a.t);
        if (a.pre_res) rd2(a.pre_res + b * a.pre_bstride, a.c, a.pre_rstride, a.t);
        if (a.post_res) rd2(a.post_res + b * a.post_bstride, a.c, a.post_rstride, a.t);
        if (a.in_mask) rd(a.in_mask + (int64_t)b * a.t, (size_t)a.t * 4);
        if (a.out_mask) rd(a.out_mask + (int64_t)b * a.t, (size_t)a.t * 4);
        wr2(a.y + b * a.y_bstride, a.c, a.y_rstride, a.t);
    }
    return TTSAMD_OK;
}
int ttsamd_rel_attention(float *out, const float *q, const float *k, const float *v, int64_t qkv_bstride, const float *mask, const float *emb_rel_k,
                         const float *emb_rel_v, int window, int batch, int heads, int dk, int t, void *)
{
    ++g_stub_launches;
    BAD(!out || !q || !k || !v || !mask || dk > 128, "rel_attention stub: bad arguments");
    for (int b = 0; b < batch; ++b) {
        rd(q + b * qkv_bstride, (size_t)heads * dk * t * 4), rd(k + b * qkv_bstride, (size_t)heads * dk * t * 4), rd(v + b * qkv_bstride, (size_t)heads * dk * t * 4);
        rd(mask + (int64_t)b * t, (size_t)t * 4);
    }
    if (emb_rel_k) rd(emb_rel_k, (size_t)(2 * window + 1) * dk * 4), rd(emb_rel_v, (size_t)(2 * window + 1) * dk * 4);
    wr2(out, 1, 0, (int64_t)batch * heads * dk * t);
    return TTSAMD_OK;
}
int ttsamd_embed(float *y, const int64_t *tokens, const float *emb, const float *mask, float, int batch, int c, int t, int vocab, void *)
{
    ++g_stub_launches;
    for (int64_t i = 0; i < (int64_t)batch * t; ++i) {
        BAD(tokens[i] < 0 || tokens[i] >= vocab, "embed stub: token id outside the table");
        rd(emb + tokens[i] * c, (size_t)c * 4);
    }
    rd(mask, (size_t)batch * t * 4);
    wr2(y, 1, 0, (int64_t)batch * c * t);
    return TTSAMD_OK;
}
int ttsamd_sequence_mask(float *mask, const int64_t *lengths, int batch, int t, void *)
{
    ++g_stub_launches;
    for (int b = 0; b < batch; ++b)
        for (int i = 0; i < t; ++i) mask[(int64_t)b * t + i] = i < lengths[b] ? 1.f : 0.f;
    return TTSAMD_OK;
}
int ttsamd_convflow_pre(float *h, const float *z, int z_ch, const float *w, const float *bias, const float *g, int batch, int c, int t, void *)
{
    ++g_stub_launches;
    BAD(z_ch < 0 || z_ch > 1, "convflow_pre stub: z channel");
    rd(z, (size_t)batch * 2 * t * 4), rd(w, (size_t)c * 4), rd(bias, (size_t)c * 4), rd(g, (size_t)batch * c * t * 4);
    wr2(h, 1, 0, (int64_t)batch * c * t);
    return TTSAMD_OK;
}
int ttsamd_convflow_spline_reverse(float *z_out, const float *z_in, const float *h, const float *mask, int batch, int t, int num_bins, float, float, void *)
{
    ++g_stub_launches;
    rd(z_in, (size_t)batch * 2 * t * 4), rd(h, (size_t)batch * (3 * num_bins - 1) * t * 4), rd(mask, (size_t)batch * t * 4);
    wr2(z_out, 1, 0, (int64_t)batch * 2 * t);
    return TTSAMD_OK;
}
int ttsamd_sdp_affine_reverse(float *z_out, const float *z_in, const float *m, const float *logs, const float *mask, int batch, int t, void *)
{
    ++g_stub_launches;
    rd(z_in, (size_t)batch * 2 * t * 4), rd(m, 8), rd(logs, 8), rd(mask, (size_t)batch * t * 4);
    wr2(z_out, 1, 0, (int64_t)batch * 2 * t);
    return TTSAMD_OK;
}
int ttsamd_durations_ex(float *durations, int32_t *cum, int64_t *y_lengths, int64_t *y_lengths_host, const float *logw, const float *durations_in,
                        const float *mask, float, int glow, int t_valid, int batch, int t, void *)
{
    ++g_stub_launches;
    BAD(!logw && !durations_in, "durations stub: neither logw nor durations_in");
    rd(logw, (size_t)batch * t * 4);
    for (int b = 0; b < batch; ++b) {
        int32_t c = 0;
        for (int i = 0; i < t; ++i) {
            float d = durations_in ? durations_in[(int64_t)b * t + i] : (mask[(int64_t)b * t + i] > 0.f ? 2.f + (i % 3) : (glow == 1 ? 1.f : 0.f));
            if (i >= t_valid) d = 0.f;
            durations[(int64_t)b * t + i] = d;
            c += (int32_t)d;
            cum[(int64_t)b * t + i] = c;
        }
        y_lengths[b] = c > 0 ? c : 1;
        if (y_lengths_host) y_lengths_host[b] = y_lengths[b];
    }
    return TTSAMD_OK;
}
int ttsamd_generate_path(float *attn, const int32_t *cum, const float *x_mask, const int64_t *y_lengths, int batch, int t_x, int t_y, void *)
{
    ++g_stub_launches;
    rd(cum, (size_t)batch * t_x * 4), rd(x_mask, (size_t)batch * t_x * 4), rd(y_lengths, (size_t)batch * 8);
    wr2(attn, 1, 0, (int64_t)batch * t_x * t_y);
    return TTSAMD_OK;
}
int ttsamd_expand_prior_ex(float *z_p, float *z_p2, float *m_p, float *logs_p, float *y_mask, const float *m, const float *logs, int64_t stats_bstride,
                           const float *noise, const int32_t *cum, const float *x_mask, const int64_t *y_lengths, float, int, int noise_packed, int batch, int c,
                           int t_x, int t_y, void *)
{
    ++g_stub_launches;
    BAD(!z_p || !y_mask || !m || !cum || !x_mask || !y_lengths, "expand_prior stub: NULL tensor");
    int64_t tmax = 0;
    for (int b = 0; b < batch; ++b) {
        rd(m + b * stats_bstride, (size_t)c * t_x * 4);
        if (logs) rd(logs + b * stats_bstride, (size_t)c * t_x * 4);
        BAD(y_lengths[b] > t_y, "expand_prior stub: y_lengths beyond the output extent");
        tmax = y_lengths[b] > tmax ? y_lengths[b] : tmax;
    }
    if (noise) rd(noise, (size_t)batch * c * (noise_packed ? tmax : t_y) * 4);
    rd(cum, (size_t)batch * t_x * 4), rd(x_mask, (size_t)batch * t_x * 4);
    const int64_t n = (int64_t)batch * c * t_y;
    wr2(z_p, 1, 0, n), wr2(z_p2, 1, 0, n), wr2(m_p, 1, 0, n), wr2(logs_p, 1, 0, n);
    for (int b = 0; b < batch; ++b)
        for (int i = 0; i < t_y; ++i) y_mask[(int64_t)b * t_y + i] = i < y_lengths[b] ? 1.f : 0.f;
    return TTSAMD_OK;
}
int ttsamd_scale(float *y, const float *x, float, int64_t n, void *)
{
    ++g_stub_launches;
    rd(x, (size_t)n * 4);
    wr2(y, 1, 0, n);
    return TTSAMD_OK;
}
int ttsamd_glow_squeeze(float *y, float *mask_out, const float *x, const float *mask, int batch, int c, int t, int n, void *)
{
    ++g_stub_launches;
    rd(x, (size_t)batch * c * t * 4), rd(mask, (size_t)batch * t * 4);
    wr2(y, 1, 0, (int64_t)batch * c * n * (t / n));
    wr2(mask_out, 1, 0, (int64_t)batch * (t / n), 1.f);
    return TTSAMD_OK;
}
int ttsamd_glow_unsqueeze(float *y, const float *x, const float *mask_q, int batch, int cq, int tq, int n, int t_out, void *)
{
    ++g_stub_launches;
    rd(x, (size_t)batch * cq * tq * 4), rd(mask_q, (size_t)batch * tq * 4);
    wr2(y, 1, 0, (int64_t)batch * (cq / n) * t_out);
    return TTSAMD_OK;
}
int ttsamd_glow_invconv_actnorm(float *x, const float *w_inv, const float *bias, const float *logs, const float *mask, int batch, int c, int t, int num_splits, int,
                                void *)
{
    ++g_stub_launches;
    BAD(num_splits != 4, "glow_invconv_actnorm stub: num_splits");
    rd(w_inv, 64), rd(bias, (size_t)c * 4), rd(logs, (size_t)c * 4), rd(mask, (size_t)batch * t * 4), rd(x, (size_t)batch * c * t * 4);
    wr2(x, 1, 0, (int64_t)batch * c * t);
    return TTSAMD_OK;
}
int ttsamd_attn_durations(float *o, const int32_t *cum, const float *x_mask, const int64_t *y_lengths, int batch, int t_x, void *)
{
    ++g_stub_launches;
    rd(cum, (size_t)batch * t_x * 4), rd(x_mask, (size_t)batch * t_x * 4), rd(y_lengths, (size_t)batch * 8);
    wr2(o, 1, 0, (int64_t)batch * t_x);
    return TTSAMD_OK;
}
int ttsamd_copy_strided(const ttsamd_copy_seg *segs, int n, void *)
{
    ++g_stub_launches;
    BAD(n < 1 || n > TTSAMD_COPY_MAX_SEGS, "copy_strided stub: segment count");
    for (int i = 0; i < n; ++i) {
        const ttsamd_copy_seg &s = segs[i];
        BAD(s.elem_bytes != 4 && s.elem_bytes != 8, "copy_strided stub: element size");
        for (int a = 0; a < s.d0; ++a)
            for (int b = 0; b < s.d1; ++b)
                for (int c = 0; c < s.d2; ++c)
                    memcpy(static_cast<char *>(s.dst) + (a * s.t0 + b * s.t1 + c * s.t2) * s.elem_bytes,
                           static_cast<const char *>(s.src) + (a * s.s0 + b * s.s1 + c * s.s2) * s.elem_bytes, (size_t)s.elem_bytes);
    }
    return TTSAMD_OK;
}

}  // extern "C"
