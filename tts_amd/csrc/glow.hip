// Glow-TTS decoder glue kernels: squeeze / unsqueeze (glow_tts/decoder.py:8-47), the 4x4 "InvConvNear"
// channel mixing (glow_tts/glow.py:107-137) fused with the ActNorm inverse that follows it in the reversed flow
// (generic/normalization.py:98-101), and the total-duration output of GlowTTS.compute_outputs
// (glow_tts.py:147).  Pure permutation / 4-wide mixing work: HBM-bound, lanes along time, one pass.
#include "common.h"

namespace ttsamd {

constexpr int kGlowThreads = 256;

// x [B,C,T] -> y [B,C*n,T/n]: y[b, s*C + c, t'] = x[b, c, t'*n + s] * mask[b, t'*n + n-1];  mask_out[b,t'] likewise
__global__ void glow_squeeze_kernel(float *__restrict__ y, float *__restrict__ mask_out, const float *__restrict__ x,
                                    const float *__restrict__ mask, int C, int T, int n, int Tq)
{
    const int b = blockIdx.z;
    const int tq = blockIdx.x * blockDim.x + threadIdx.x;
    if (tq >= Tq) return;
    const float m = mask ? mask[(long)b * T + tq * n + n - 1] : 1.f;
    if (blockIdx.y == 0 && mask_out) mask_out[(long)b * Tq + tq] = m;
    for (int cc = blockIdx.y; cc < C * n; cc += gridDim.y) {
        const int s = cc / C, c = cc - s * C;
        y[((long)b * C * n + cc) * Tq + tq] = x[((long)b * C + c) * T + (long)tq * n + s] * m;
    }
}

// x [B,Cq,Tq] -> y [B,Cq/n,T_out]: y[b, c, t'*n + s] = x[b, s*(Cq/n) + c, t'] * mask_q[b,t'] ; columns >= Tq*n are zero
__global__ void glow_unsqueeze_kernel(float *__restrict__ y, const float *__restrict__ x,
                                      const float *__restrict__ mask_q, int Cq, int Tq, int n, int T_out)
{
    const int b = blockIdx.z;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T_out) return;
    const int tq = t / n, s = t - tq * n;
    const int C = Cq / n;
    const bool in = tq < Tq;
    const float m = (in && mask_q) ? mask_q[(long)b * Tq + tq] : 1.f;
    for (int c = blockIdx.y; c < C; c += gridDim.y)
        y[((long)b * C + c) * T_out + t] = in ? x[((long)b * Cq + s * C + c) * Tq + tq] * m : 0.f;
}

// In place on x [B,C,T] (C = 2 * (C/ns) * (ns/2) grouping of glow.py:116-117):
//   group g = a*(ns/2) + d  <->  channel a*(C/2) + q*(ns/2) + d,  q < C/ns
//   z[g'] = sum_g w_inv[g'][g] * x[g];  z *= mask;  z = (z - bias[ch]) * exp(-logs[ch]) * mask   (ActNorm reverse)
// forward == 0 (reverse flow): z = (W x) * mask;  x = (z - bias) * exp(-logs) * mask      (W = stored inverse)
// forward == 1 (forward flow): x = (bias + exp(logs) * x) * mask;  x = (W x) * mask           (W = the weight itself)
template <int NS>
__global__ void glow_invconv_actnorm_kernel(float *__restrict__ x, const float *__restrict__ w_inv,
                                            const float *__restrict__ bias, const float *__restrict__ logs,
                                            const float *__restrict__ mask, int C, int T, int forward)
{
    const int b = blockIdx.z;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const float m = mask ? mask[(long)b * T + t] : 1.f;
    float w[NS][NS];
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int jx = 0; jx < NS; ++jx) w[i][jx] = w_inv[i * NS + jx];
    const int Q = C / NS;
    for (int q = blockIdx.y; q < Q; q += gridDim.y) {
        float v[NS];
        long off[NS];
#pragma unroll
        for (int g = 0; g < NS; ++g) {
            const int a = g / (NS / 2), d = g - a * (NS / 2);
            const int ch = a * (C / 2) + q * (NS / 2) + d;
            off[g] = ((long)b * C + ch) * T + t;
            v[g] = x[off[g]];
            if (forward && bias) v[g] = (bias[ch] + expf(logs[ch]) * v[g]) * m;
        }
#pragma unroll
        for (int go = 0; go < NS; ++go) {
            float z = 0.f;
#pragma unroll
            for (int g = 0; g < NS; ++g) z += w[go][g] * v[g];
            const int a = go / (NS / 2), d = go - a * (NS / 2);
            const int ch = a * (C / 2) + q * (NS / 2) + d;
            z *= m;
            if (!forward && bias) z = (z - bias[ch]) * expf(-logs[ch]) * m;
            x[off[go]] = z;
        }
    }
}

// o[b,x] = log(1 + sum_y attn[b,x,y]) * x_mask[b,x], with the row sum taken from the cumulative durations
__global__ void attn_durations_kernel(float *__restrict__ o, const int *__restrict__ cum,
                                      const float *__restrict__ x_mask, const long *__restrict__ y_lengths, int Tx)
{
    const int b = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= Tx) return;
    const long yl = y_lengths[b];
    const long hi = min((long)cum[(long)b * Tx + x], yl);
    const long lo = x > 0 ? min((long)cum[(long)b * Tx + x - 1], yl) : 0;
    const float xm = x_mask ? x_mask[(long)b * Tx + x] : 1.f;
    const float s = (float)(hi - lo) * xm;
    o[(long)b * Tx + x] = logf(1.f + s) * xm;
}

// o[r] = sum_t x[r, t]  (durations of a MAS alignment: attn.sum(-1)); one wavefront per row
__global__ void row_sum_kernel(float *__restrict__ o, const float *__restrict__ x, int t)
{
    const long r = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < t; i += 64) s += x[r * t + i];
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) s += __shfl_xor(s, k);
    if (threadIdx.x == 0) o[r] = s;
}

}  // namespace ttsamd
using namespace ttsamd;

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

extern "C" int ttsamd_glow_squeeze(float *y, float *mask_out, const float *x, const float *mask, int batch, int c,
                                   int t, int n, void *stream)
{
    TTSAMD_CHECK_ARG(y && x && batch >= 0 && c > 0 && t >= 0 && n >= 1, "glow_squeeze: bad args");
    const int tq = t / n;
    if (batch == 0 || tq == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(batch <= 65535, "glow_squeeze: batch > 65535");
    hipLaunchKernelGGL(glow_squeeze_kernel, dim3(cdiv(tq, 64), min(c * n, 32), batch), dim3(64), 0, as_stream(stream), y,
                       mask_out, x, mask, c, t, n, tq);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_glow_unsqueeze(float *y, const float *x, const float *mask_q, int batch, int cq, int tq, int n,
                                     int t_out, void *stream)
{
    TTSAMD_CHECK_ARG(y && x && batch >= 0 && cq > 0 && tq >= 0 && n >= 1 && cq % n == 0 && t_out >= tq * n,
                     "glow_unsqueeze: bad args");
    if (batch == 0 || t_out == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(batch <= 65535, "glow_unsqueeze: batch > 65535");
    hipLaunchKernelGGL(glow_unsqueeze_kernel, dim3(cdiv(t_out, 64), min(cq / n, 32), batch), dim3(64), 0,
                       as_stream(stream), y, x, mask_q, cq, tq, n, t_out);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_glow_invconv_actnorm(float *x, const float *w_inv, const float *bias, const float *logs,
                                           const float *mask, int batch, int c, int t, int num_splits, int forward,
                                           void *stream)
{
    TTSAMD_CHECK_ARG(x && w_inv && batch >= 0 && c > 0 && t >= 0, "glow_invconv_actnorm: bad args");
    TTSAMD_CHECK_ARG((bias == nullptr) == (logs == nullptr), "glow_invconv_actnorm: need both or neither of bias/logs");
    if (num_splits != 4 || c % 4 != 0) {
        set_error("glow_invconv_actnorm: only num_splits == 4 (GlowTTSConfig default) is built, C %% 4 == 0");
        return TTSAMD_ERR_UNSUPPORTED;
    }
    if (batch == 0 || t == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(batch <= 65535, "glow_invconv_actnorm: batch > 65535");
    hipLaunchKernelGGL(glow_invconv_actnorm_kernel<4>, dim3(cdiv(t, 64), min(c / 4, 16), batch), dim3(64), 0,
                       as_stream(stream), x, w_inv, bias, logs, mask, c, t, forward);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_attn_durations(float *o, const int32_t *cum, const float *x_mask, const int64_t *y_lengths,
                                     int batch, int t_x, void *stream)
{
    TTSAMD_CHECK_ARG(o && cum && y_lengths && batch >= 0 && t_x >= 0, "attn_durations: bad args");
    if (batch == 0 || t_x == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(batch <= 65535, "attn_durations: batch > 65535");
    hipLaunchKernelGGL(attn_durations_kernel, dim3(cdiv(t_x, kGlowThreads), batch), dim3(kGlowThreads), 0,
                       as_stream(stream), o, cum, x_mask, reinterpret_cast<const long *>(y_lengths), t_x);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_row_sum(float *o, const float *x, int64_t rows, int t, void *stream)
{
    TTSAMD_CHECK_ARG(o && x && rows >= 0 && t >= 0 && rows <= 0x7FFFFFFF, "row_sum: bad args");
    if (rows == 0) return TTSAMD_OK;
    hipLaunchKernelGGL(row_sum_kernel, dim3((unsigned)rows), dim3(64), 0, as_stream(stream), o, x, t);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}
