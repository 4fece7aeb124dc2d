// Text-side small kernels of the VITS / Glow-TTS inference path: embedding, sequence masks, the
// stochastic duration predictor's flow tails (rational-quadratic spline inverse), durations ->
// cumulative frame offsets, generate_path and the prior expansion (declarations + reference
// citations: include/tts_amd.h).  All HBM/latency-bound: lanes run along time (the contiguous axis
// of the channels-first layout), one pass over each tensor, no reshaping into GEMMs.
#include "common.h"

namespace ttsamd {

constexpr int kTxtThreads = 256;

// ---- embedding ------------------------------------------------------------------------------
__global__ void embed_kernel(float *__restrict__ y, const long *__restrict__ tokens, const float *__restrict__ emb,
                             const float *__restrict__ mask, float scale, int C, int T, int V,
                             const float *__restrict__ extra, int CE)
{
    const int b = blockIdx.z;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    long tok = tokens[(long)b * T + t];
    tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
    const float m = mask ? mask[(long)b * T + t] : 1.f;
    for (int c = blockIdx.y; c < C; c += gridDim.y) {
        float v = emb[tok * C + c] * scale;
        if (mask) v *= m;
        y[((long)b * (C + CE) + c) * T + t] = v;
    }
    for (int c = blockIdx.y; c < CE; c += gridDim.y) {   // per-item channels appended unscaled (language embedding)
        float v = extra[(long)b * CE + c];
        if (mask) v *= m;
        y[((long)b * (C + CE) + C + c) * T + t] = v;
    }
}

__global__ void sequence_mask_kernel(float *__restrict__ mask, const long *__restrict__ lengths, int T)
{
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) mask[(long)b * T + t] = (t < lengths[b]) ? 1.f : 0.f;
}

// ---- ConvFlow head --------------------------------------------------------------------------
__global__ void convflow_pre_kernel(float *__restrict__ h, const float *__restrict__ z, int z_ch,
                                    const float *__restrict__ w, const float *__restrict__ bias,
                                    const float *__restrict__ g, int C, int T)
{
    const int b = blockIdx.z;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const float x0 = z[((long)b * 2 + z_ch) * T + t];
    // four channels per pass with their loads in flight together (one channel per pass was one memory round trip per channel:
    // twelve in a row at C = 192; the launch is a handful of blocks, so its time is that chain)
    for (int c0 = blockIdx.y; c0 < C; c0 += 4 * gridDim.y) {
        float wv[4], bv[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * gridDim.y;
            const int cc = c < C ? c : c0;                      // a pass's tail re-reads its first channel (never stored)
            wv[u] = w[cc];
            bv[u] = bias ? bias[cc] : 0.f;
            gv[u] = g ? g[((long)b * C + cc) * T + t] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * gridDim.y;
            float v = wv[u] * x0;
            if (bias) v += bv[u];
            if (g) v += gv[u];
            if (c < C) h[((long)b * C + c) * T + t] = v;
        }
    }
}

// ---- rational-quadratic spline, inverse direction, linear tails --------------------------------
// Follows vits/transforms.py:50-184 operation by operation (fp32, no contraction).
constexpr int kMaxBins = 16;
constexpr float kMinBinWidth = 1e-3f, kMinBinHeight = 1e-3f, kMinDerivative = 1e-3f;

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// knots[0..nb] from unnormalised params (softmax -> min width -> cumsum -> affine to [-B, B])
__device__ __forceinline__ void spline_knots(float *knots, float *sizes, const float *u, int nb, float min_size,
                                             float lo, float hi)
{
    float mx = u[0];
    for (int k = 1; k < nb; ++k) mx = fmaxf(mx, u[k]);
    float e[kMaxBins];
    float sum = 0.f;
    for (int k = 0; k < nb; ++k) {
        e[k] = expf(u[k] - mx);
        sum += e[k];
    }
    const float keep = (float)(1.0 - (double)min_size * nb);  // python double, then fp32 multiply
    float cum = 0.f;
    knots[0] = lo;
    for (int k = 0; k < nb; ++k) {
        const float wk = min_size + keep * (e[k] / sum);
        cum += wk;
        knots[k + 1] = (hi - lo) * cum + lo;
    }
    knots[nb] = hi;
    for (int k = 0; k < nb; ++k) sizes[k] = knots[k + 1] - knots[k];
}

// NBT > 0: the bin count as a compile-time constant (VITS: 10) — every loop unrolls, every array lives in registers (the generic
// form keeps 224 bytes of scratch per thread and walks its 2 nb + 2 parameter loads as a chain of dependent round trips: 15 us per
// launch for five blocks of work), all 3 nb - 1 parameters are requested together, and the bin's entries are picked by selects.
// Same operations in the same order on the same operands: bitwise the generic kernel.
template <int NBT>
__global__ void convflow_spline_reverse_kernel(float *__restrict__ z_out, const float *__restrict__ z_in,
                                               const float *__restrict__ h, const float *__restrict__ mask,
                                               int T, int nb_arg, float sqrt_filter, float tail)
{
    const int nb = NBT > 0 ? NBT : nb_arg;
    constexpr int NA = NBT > 0 ? NBT : kMaxBins;
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const float m = mask ? mask[(long)b * T + t] : 1.f;
    const float x0 = z_in[((long)b * 2 + 1) * T + t];
    const float x1 = z_in[((long)b * 2 + 0) * T + t];
    float outv = x1;
    if (x1 >= -tail && x1 <= tail) {
        const float *hp = h + (long)b * (3 * nb - 1) * T + t;
        float uw[NA], uh[NA], cw[NA + 1], ch[NA + 1], wd[NA], ht[NA];
        float rd[NA];                                 // unnormalised derivatives of the interior knots (NBT > 0 only)
        if constexpr (NBT > 0) {
            float rw[NA], rh[NA];
#pragma unroll
            for (int k = 0; k < NBT; ++k) {
                rw[k] = hp[(long)k * T];
                rh[k] = hp[(long)(NBT + k) * T];
                rd[k] = (k < NBT - 1) ? hp[(long)(2 * NBT + k) * T] : 0.f;      // (compile-time condition)
            }
#pragma unroll
            for (int k = 0; k < NBT; ++k) {
                uw[k] = rw[k] / sqrt_filter;
                uh[k] = rh[k] / sqrt_filter;
            }
        } else {
            for (int k = 0; k < nb; ++k) {
                uw[k] = hp[(long)k * T] / sqrt_filter;
                uh[k] = hp[(long)(nb + k) * T] / sqrt_filter;
            }
        }
        spline_knots(cw, wd, uw, nb, kMinBinWidth, -tail, tail);
        spline_knots(ch, ht, uh, nb, kMinBinHeight, -tail, tail);
        // searchsorted(cumheights, x): count of knots <= x (last knot + 1e-6), minus one
        int bin = -1;
#pragma unroll
        for (int k = 0; k <= NA; ++k) {
            if (k <= nb) {
                const float loc = (k == nb) ? ch[k] + 1e-6f : ch[k];
                bin += (x1 >= loc) ? 1 : 0;
            }
        }
        bin = bin < 0 ? 0 : (bin > nb - 1 ? nb - 1 : bin);
        const float edge = (float)log(exp(1.0 - (double)kMinDerivative) - 1.0);
        float ud0, ud1, in_cw, in_w, in_ch, in_h;
        if constexpr (NBT > 0) {
            ud0 = edge;
            ud1 = edge;
            in_cw = cw[0], in_w = wd[0], in_ch = ch[0], in_h = ht[0];
#pragma unroll
            for (int k = 0; k < NBT; ++k) {
                if (k >= 1) {
                    ud0 = (bin == k) ? rd[k - 1] : ud0;
                    in_cw = (bin == k) ? cw[k] : in_cw;
                    in_w = (bin == k) ? wd[k] : in_w;
                    in_ch = (bin == k) ? ch[k] : in_ch;
                    in_h = (bin == k) ? ht[k] : in_h;
                }
                if (k < NBT - 1) ud1 = (bin == k) ? rd[k] : ud1;
            }
        } else {
            ud0 = (bin == 0) ? edge : hp[(long)(2 * nb + bin - 1) * T];
            ud1 = (bin == nb - 1) ? edge : hp[(long)(2 * nb + bin) * T];
            in_cw = cw[bin], in_w = wd[bin], in_ch = ch[bin], in_h = ht[bin];
        }
        const float d0 = kMinDerivative + softplus_f(ud0);
        const float d1 = kMinDerivative + softplus_f(ud1);
        const float delta = in_h / in_w;
        const float dx = x1 - in_ch;
        const float s = d0 + d1 - 2.f * delta;
        const float qa = dx * s + in_h * (delta - d0);
        const float qb = in_h * d0 - dx * s;
        const float qc = -delta * dx;
        const float disc = qb * qb - 4.f * qa * qc;
        const float root = (2.f * qc) / (-qb - sqrtf(disc));
        outv = root * in_w + in_cw;
    }
    z_out[((long)b * 2 + 0) * T + t] = x0 * m;
    z_out[((long)b * 2 + 1) * T + t] = outv * m;
}

__global__ void sdp_affine_reverse_kernel(float *__restrict__ z_out, const float *__restrict__ z_in,
                                          const float *__restrict__ mm, const float *__restrict__ logs,
                                          const float *__restrict__ mask, int T)
{
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const float m = mask ? mask[(long)b * T + t] : 1.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float zf = z_in[((long)b * 2 + (1 - c)) * T + t];
        z_out[((long)b * 2 + c) * T + t] = (zf - mm[c]) * expf(-logs[c]) * m;
    }
}

// ---- durations -> cumulative frame offsets -----------------------------------------------------
__global__ __launch_bounds__(kTxtThreads) void durations_kernel(float *__restrict__ dur, int *__restrict__ cum,
                                                                long *__restrict__ y_lengths,
                                                                long *__restrict__ y_lengths_host,
                                                                const float *__restrict__ logw,
                                                                const float *__restrict__ dur_in,
                                                                const float *__restrict__ mask, float length_scale,
                                                                int glow, int t_valid, int T)
{
    __shared__ int part[kTxtThreads];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int per = (T + kTxtThreads - 1) / kTxtThreads;
    const int lo = tid * per;
    const int hi = min(T, lo + per);
    int local = 0;
    for (int t = lo; t < hi; ++t) {
        const long o = (long)b * T + t;
        float wc;
        if (dur_in) {
            wc = dur_in[o];
        } else {
            const float m = mask ? mask[o] : 1.f;
            float w = expf(logw[o]);
            if (glow) w = w - 1.f;
            w = w * m * length_scale;
            wc = ceilf(w);
            if (glow) wc = fmaxf(wc, 1.f);
            if (glow == 2) wc *= m;   // ragged-exact batching: padded tokens own no frame
        }
        if (t >= t_valid) wc = 0.f;   // columns of a text-length bucket beyond the caller's tensor: no frames, whatever the rule
        dur[o] = wc;
        const float cl = fminf(fmaxf(wc, 0.f), 1048576.f);  // keep the int offsets finite for inf/NaN inputs
        local += (int)cl;
    }
    part[tid] = local;
    __syncthreads();
    // inclusive scan over the 256 partials (Hillis-Steele; tiny)
    for (int off = 1; off < kTxtThreads; off <<= 1) {
        const int v = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = (tid > 0) ? part[tid - 1] : 0;
    for (int t = lo; t < hi; ++t) {
        const long o = (long)b * T + t;
        const float cl = fminf(fmaxf(dur[o], 0.f), 1048576.f);
        run += (int)cl;
        cum[o] = run;
    }
    if (tid == kTxtThreads - 1) {
        const int tot = part[kTxtThreads - 1];
        const long yl = tot < 1 ? 1 : tot;
        y_lengths[b] = yl;
        if (y_lengths_host) {          // mirror in host-mapped (pinned) memory: the host polls it instead of a stream sync + D2H
            __hip_atomic_store(y_lengths_host + b, yl, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ void generate_path_kernel(float *__restrict__ attn, const int *__restrict__ cum,
                                     const float *__restrict__ x_mask, const long *__restrict__ y_lengths, int Tx,
                                     int Ty)
{
    const int b = blockIdx.z;
    const int x = blockIdx.y;
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= Ty) return;
    const int hi = cum[(long)b * Tx + x];
    const int lo = x > 0 ? cum[(long)b * Tx + x - 1] : 0;
    const float xm = x_mask ? x_mask[(long)b * Tx + x] : 1.f;
    const float on = (y >= lo && y < hi && y < y_lengths[b]) ? xm : 0.f;
    attn[((long)b * Tx + x) * Ty + y] = on;
}

__global__ __launch_bounds__(256) void expand_prior_kernel(
    float *__restrict__ z_p, float *__restrict__ z_p2, float *__restrict__ m_p, float *__restrict__ logs_p,
    float *__restrict__ y_mask, const float *__restrict__ m, const float *__restrict__ logs, long stats_bstride,
    const float *__restrict__ noise, const int *__restrict__ cum, const float *__restrict__ x_mask,
    const long *__restrict__ y_lengths, float noise_scale, int mask_out, int noise_packed, int C, int Tx, int Ty)
{
    const int b = blockIdx.y;
    long tn = Ty;                     // extent (and row stride) of the noise tensor
    if (noise && noise_packed) {      // a contiguous [B, C, max(y_lengths)] draw: beyond that extent the noise is zero
        tn = 1;
        for (int i = 0; i < (int)gridDim.y; ++i) tn = y_lengths[i] > tn ? y_lengths[i] : tn;    // gridDim.y = batch
    }
    const int y = blockIdx.x * 64 + threadIdx.x;
    const int grp = threadIdx.y;
    if (y >= Ty) return;
    const int *cb = cum + (long)b * Tx;
    // first x with cum[x] > y
    int lo = 0, hi = Tx;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cb[mid] > y) hi = mid; else lo = mid + 1;
    }
    const int x = lo;
    const bool in_len = y < y_lengths[b];
    const bool valid = in_len && x < Tx && (!x_mask || x_mask[(long)b * Tx + x] != 0.f);
    const float ym = in_len ? 1.f : 0.f;
    // channels are cut into slices of 16 along blockIdx.z (a single request launches 13 column tiles: one block per tile
    // walked its 48 channel rounds alone for 50 us)
    const int c_lo = blockIdx.z * 16, c_hi = min(C, c_lo + 16);
    if (grp == 0 && y_mask && blockIdx.z == 0) y_mask[(long)b * Ty + y] = ym;
    for (int c = c_lo + grp; c < c_hi; c += 4) {
        const long src = (long)b * stats_bstride + (long)c * Tx + x;
        const long dst = ((long)b * C + c) * Ty + y;
        const float mv = valid ? m[src] : 0.f;
        const float lv = (valid && logs) ? logs[src] : 0.f;
        float z = mv;
        if (noise) {
            const float nv = noise_packed ? (y < tn ? noise[((long)b * C + c) * tn + y] : 0.f) : noise[dst];
            z = mv + nv * expf(lv) * noise_scale;
        }
        if (mask_out) z *= ym;
        if (m_p) m_p[dst] = mv;
        if (logs_p) logs_p[dst] = lv;
        z_p[dst] = z;
        if (z_p2) z_p2[dst] = z;
    }
}

__global__ void scale_kernel(float *__restrict__ y, const float *__restrict__ x, float s, long n)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = x[i] * s;
}

// y = ((a + b) [+ c]) / div: the MRF's z_sum / num_kernels over separately written branch outputs, in the reference's order
// (hifigan_generator.py:255-261)
__global__ void sum_div_kernel(float *__restrict__ y, const float *__restrict__ a, const float *__restrict__ b,
                               const float *__restrict__ c, float div, long n4)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 va = reinterpret_cast<const float4 *>(a)[i], vb = reinterpret_cast<const float4 *>(b)[i];
        float4 v{va.x + vb.x, va.y + vb.y, va.z + vb.z, va.w + vb.w};
        if (c) {
            const float4 vc = reinterpret_cast<const float4 *>(c)[i];
            v = float4{v.x + vc.x, v.y + vc.y, v.z + vc.z, v.w + vc.w};
        }
        reinterpret_cast<float4 *>(y)[i] = float4{v.x / div, v.y / div, v.z / div, v.w / div};
    }
}

// z[b,c,t] = (m[b,c,t] + noise[b,c,t] * exp(logs[b,c,t])) * mask[b,t]; m / logs are the halves of one [B,2C,T] buffer
__global__ void sample_gaussian_kernel(float *__restrict__ z, const float *__restrict__ stats,
                                       const float *__restrict__ noise, const float *__restrict__ mask, int C, int T)
{
    const int b = blockIdx.z;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const float mk = mask ? mask[(long)b * T + t] : 1.f;
    for (int c = blockIdx.y; c < C; c += gridDim.y) {
        const float mv = stats[((long)b * 2 * C + c) * T + t];
        const float lv = stats[((long)b * 2 * C + C + c) * T + t];
        const long o = ((long)b * C + c) * T + t;
        z[o] = (mv + noise[o] * expf(lv)) * mk;
    }
}

}  // namespace ttsamd
using namespace ttsamd;

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

extern "C" int ttsamd_embed(float *y, const int64_t *tokens, const float *emb, const float *mask, float scale,
                            int batch, int c, int t, int vocab, void *stream)
{
    TTSAMD_CHECK_ARG(y && tokens && emb && batch >= 0 && c > 0 && t >= 0 && vocab > 0, "embed: bad args");
    if (batch == 0 || t == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(batch <= 65535, "embed: batch > 65535");
    hipLaunchKernelGGL(embed_kernel, dim3(cdiv(t, 64), min(c, 16), batch), dim3(64), 0, as_stream(stream), y,
                       reinterpret_cast<const long *>(tokens), emb, mask, scale, c, t, vocab, nullptr, 0);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_embed_cat(float *y, const int64_t *tokens, const float *emb, const float *mask, float scale,
                                const float *extra, int c_extra, int batch, int c, int t, int vocab, void *stream)
{
    TTSAMD_CHECK_ARG(y && tokens && emb && batch >= 0 && c > 0 && t >= 0 && vocab > 0, "embed_cat: bad args");
    TTSAMD_CHECK_ARG(c_extra >= 0 && (c_extra == 0 || extra), "embed_cat: extra channels without a table");
    if (batch == 0 || t == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(batch <= 65535, "embed_cat: batch > 65535");
    hipLaunchKernelGGL(embed_kernel, dim3(cdiv(t, 64), min(c, 16), batch), dim3(64), 0, as_stream(stream), y,
                       reinterpret_cast<const long *>(tokens), emb, mask, scale, c, t, vocab, extra, c_extra);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_sequence_mask(float *mask, const int64_t *lengths, int batch, int t, void *stream)
{
    TTSAMD_CHECK_ARG(mask && lengths && batch >= 0 && t >= 0, "sequence_mask: bad args");
    if (batch == 0 || t == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(batch <= 65535, "sequence_mask: batch > 65535");
    hipLaunchKernelGGL(sequence_mask_kernel, dim3(cdiv(t, kTxtThreads), batch), dim3(kTxtThreads), 0,
                       as_stream(stream), mask, reinterpret_cast<const long *>(lengths), t);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_convflow_pre(float *h, const float *z, int z_ch, const float *w, const float *bias,
                                   const float *g, int batch, int c, int t, void *stream)
{
    TTSAMD_CHECK_ARG(h && z && w && (z_ch == 0 || z_ch == 1) && batch >= 0 && c > 0 && t >= 0, "convflow_pre: bad args");
    if (batch == 0 || t == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(batch <= 65535, "convflow_pre: batch > 65535");
    hipLaunchKernelGGL(convflow_pre_kernel, dim3(cdiv(t, 64), min(c, 48), batch), dim3(64), 0, as_stream(stream), h, z,
                       z_ch, w, bias, g, c, t);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_convflow_spline_reverse(float *z_out, const float *z_in, const float *h, const float *mask,
                                              int batch, int t, int num_bins, float filter_channels,
                                              float tail_bound, void *stream)
{
    TTSAMD_CHECK_ARG(z_out && z_in && h && batch >= 0 && t >= 0 && filter_channels > 0 && tail_bound > 0,
                     "convflow_spline_reverse: bad args");
    TTSAMD_CHECK_ARG(z_out != z_in, "convflow_spline_reverse: in-place not allowed (channel flip)");
    if (num_bins < 2 || num_bins > kMaxBins) {
        set_error("convflow_spline_reverse: num_bins=%d outside [2,%d]", num_bins, kMaxBins);
        return TTSAMD_ERR_UNSUPPORTED;
    }
    if (batch == 0 || t == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(batch <= 65535, "convflow_spline_reverse: batch > 65535");
    if (num_bins == 10)      // VITS's stochastic duration predictor (stochastic_duration_predictor.py: num_bins = 10)
        hipLaunchKernelGGL(convflow_spline_reverse_kernel<10>, dim3(cdiv(t, 64), batch), dim3(64), 0, as_stream(stream), z_out,
                           z_in, h, mask, t, num_bins, sqrtf(filter_channels), tail_bound);
    else
        hipLaunchKernelGGL(convflow_spline_reverse_kernel<0>, dim3(cdiv(t, 64), batch), dim3(64), 0, as_stream(stream), z_out,
                           z_in, h, mask, t, num_bins, sqrtf(filter_channels), tail_bound);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_sdp_affine_reverse(float *z_out, const float *z_in, const float *m, const float *logs,
                                         const float *mask, int batch, int t, void *stream)
{
    TTSAMD_CHECK_ARG(z_out && z_in && m && logs && batch >= 0 && t >= 0, "sdp_affine_reverse: bad args");
    TTSAMD_CHECK_ARG(z_out != z_in, "sdp_affine_reverse: in-place not allowed (channel flip)");
    if (batch == 0 || t == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(batch <= 65535, "sdp_affine_reverse: batch > 65535");
    hipLaunchKernelGGL(sdp_affine_reverse_kernel, dim3(cdiv(t, kTxtThreads), batch), dim3(kTxtThreads), 0,
                       as_stream(stream), z_out, z_in, m, logs, mask, t);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_durations(float *durations, int32_t *cum, int64_t *y_lengths, const float *logw,
                                const float *durations_in, const float *mask, float length_scale, int glow,
                                int batch, int t, void *stream)
{
    TTSAMD_CHECK_ARG(durations && cum && y_lengths && (logw || durations_in) && batch >= 0 && t > 0,
                     "durations: bad args");
    if (batch == 0) return TTSAMD_OK;
    hipLaunchKernelGGL(durations_kernel, dim3(batch), dim3(kTxtThreads), 0, as_stream(stream), durations, cum,
                       reinterpret_cast<long *>(y_lengths), (long *)nullptr, logw, durations_in, mask, length_scale, glow, t, t);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_durations_ex(float *durations, int32_t *cum, int64_t *y_lengths, int64_t *y_lengths_host,
                                   const float *logw, const float *durations_in, const float *mask, float length_scale,
                                   int glow, int t_valid, int batch, int t, void *stream)
{
    TTSAMD_CHECK_ARG(durations && cum && y_lengths && (logw || durations_in) && batch >= 0 && t > 0 && t_valid >= 0,
                     "durations_ex: bad args");
    if (batch == 0) return TTSAMD_OK;
    hipLaunchKernelGGL(durations_kernel, dim3(batch), dim3(kTxtThreads), 0, as_stream(stream), durations, cum,
                       reinterpret_cast<long *>(y_lengths), reinterpret_cast<long *>(y_lengths_host), logw, durations_in, mask,
                       length_scale, glow, t_valid < t ? t_valid : t, t);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_generate_path(float *attn, const int32_t *cum, const float *x_mask, const int64_t *y_lengths,
                                    int batch, int t_x, int t_y, void *stream)
{
    TTSAMD_CHECK_ARG(attn && cum && y_lengths && batch >= 0 && t_x >= 0 && t_y >= 0, "generate_path: bad args");
    if (batch == 0 || t_x == 0 || t_y == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(batch <= 65535 && t_x <= 65535, "generate_path: batch / t_x > 65535");
    hipLaunchKernelGGL(generate_path_kernel, dim3(cdiv(t_y, kTxtThreads), t_x, batch), dim3(kTxtThreads), 0,
                       as_stream(stream), attn, cum, x_mask, reinterpret_cast<const long *>(y_lengths), t_x, t_y);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_expand_prior(float *z_p, float *z_p2, float *m_p, float *logs_p, float *y_mask, const float *m,
                                   const float *logs, int64_t stats_bstride, const float *noise, const int32_t *cum,
                                   const float *x_mask, const int64_t *y_lengths, float noise_scale, int mask_out,
                                   int batch, int c, int t_x, int t_y, void *stream)
{
    return ttsamd_expand_prior_ex(z_p, z_p2, m_p, logs_p, y_mask, m, logs, stats_bstride, noise, cum, x_mask, y_lengths,
                                  noise_scale, mask_out, 0, batch, c, t_x, t_y, stream);
}

extern "C" int ttsamd_expand_prior_ex(float *z_p, float *z_p2, float *m_p, float *logs_p, float *y_mask, const float *m,
                                      const float *logs, int64_t stats_bstride, const float *noise, const int32_t *cum,
                                      const float *x_mask, const int64_t *y_lengths, float noise_scale, int mask_out,
                                      int noise_packed, int batch, int c, int t_x, int t_y, void *stream)
{
    TTSAMD_CHECK_ARG(z_p && m && cum && y_lengths && batch >= 0 && c > 0 && t_x > 0 && t_y >= 0,
                     "expand_prior: bad args");
    if (batch == 0 || t_y == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(batch <= 65535, "expand_prior: batch > 65535");
    hipLaunchKernelGGL(expand_prior_kernel, dim3(cdiv(t_y, 64), batch, cdiv(c, 16)), dim3(64, 4), 0, as_stream(stream), z_p, z_p2,
                       m_p, logs_p, y_mask, m, logs, (long)stats_bstride, noise, cum, x_mask,
                       reinterpret_cast<const long *>(y_lengths),
                       noise_scale, mask_out, noise_packed, c, t_x, t_y);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_scale(float *y, const float *x, float s, int64_t n, void *stream)
{
    TTSAMD_CHECK_ARG(y && x && n >= 0, "scale: bad args");
    if (n == 0) return TTSAMD_OK;
    const long blocks = (n + kTxtThreads - 1) / kTxtThreads;
    hipLaunchKernelGGL(scale_kernel, dim3((int)(blocks > 8192 ? 8192 : blocks)), dim3(kTxtThreads), 0, as_stream(stream),
                       y, x, s, (long)n);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_sum_div(float *y, const float *a, const float *b, const float *c, float div, int64_t n, void *stream)
{
    TTSAMD_CHECK_ARG(y && a && b && n >= 0 && div != 0.f, "sum_div: bad args");
    TTSAMD_CHECK_ARG(n % 4 == 0 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) |
                                     reinterpret_cast<uintptr_t>(c)) & 15) == 0, "sum_div: 16-byte aligned tensors of a multiple of 4 elements");
    if (n == 0) return TTSAMD_OK;
    const long n4 = n / 4, blocks = (n4 + kTxtThreads - 1) / kTxtThreads;
    hipLaunchKernelGGL(sum_div_kernel, dim3((int)(blocks > 8192 ? 8192 : blocks)), dim3(kTxtThreads), 0, as_stream(stream), y, a, b, c,
                       div, n4);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_sample_gaussian(float *z, const float *stats, const float *noise, const float *mask, int batch, int c,
                                      int t, void *stream)
{
    TTSAMD_CHECK_ARG(z && stats && noise && batch >= 0 && c > 0 && t >= 0, "sample_gaussian: bad args");
    if (batch == 0 || t == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(batch <= 65535, "sample_gaussian: batch > 65535");
    hipLaunchKernelGGL(sample_gaussian_kernel, dim3(cdiv(t, 64), min(c, 32), batch), dim3(64), 0, as_stream(stream), z, stats,
                       noise, mask, c, t);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}
