// Channel LayerNorm on channels-first [B, C, T] tensors with fused depthwise-conv prologue,
// residual adds, activation and masking (see include/tts_amd.h: ttsamd_channel_norm).
//
// Replaces generic/normalization.py:5-53 (LayerNorm / LayerNorm2) and the per-layer chain of
// DilatedDepthSeparableConv (vits/stochastic_duration_predictor.py:46-63).
//
// HBM/latency-bound.  Layout decision: the normalised axis (C) is the STRIDED one, so lanes run along
// time (coalesced 256-byte row segments per wavefront) and each workgroup is 64 time columns x 16 channel
// groups; a thread keeps its C/16 channel values in registers between the statistics passes (the tensor
// is read exactly once), partial sums meet in LDS in a fixed order (deterministic).
#include "common.h"

namespace ttsamd {

constexpr int kNormSmallT = 2048;   // up to this many columns the 16-column tiles are used

template <int NC, int kNormGroups>  // channels per thread, channel groups per block (C <= NC * kNormGroups)
__global__ __launch_bounds__(64 * kNormGroups) void channel_norm_kernel(const ttsamd_norm_args a)
{
    __shared__ float red[kNormGroups][64];
    const int lane = threadIdx.x;       // time lane
    const int grp = threadIdx.y;        // channel group: channels grp, grp+4, ...
    const int b = blockIdx.y;
    const int t = blockIdx.x * 64 + lane;
    const bool tv = t < a.t;
    const float *xb = a.x + (long)b * a.x_bstride;
    const float *im = a.in_mask ? a.in_mask + (long)b * a.t : nullptr;

    float v[NC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = grp + i * kNormGroups;
        float u = 0.f;
        if (tv && c < a.c) {
            if (a.dw_w) {
                u = a.dw_bias ? a.dw_bias[c] : 0.f;
                const int half = (a.dw_kernel - 1) / 2;
                for (int k = 0; k < a.dw_kernel; ++k) {
                    const int tt = t + (k - half) * a.dw_dilation;
                    if (tt >= 0 && tt < a.t) {
                        float xv = xb[(long)c * a.x_rstride + tt];
                        if (im) xv *= im[tt];
                        u += a.dw_w[c * a.dw_kernel + k] * xv;
                    }
                }
            } else {
                u = xb[(long)c * a.x_rstride + t];
            }
            if (a.pre_res) u += a.pre_res[(long)b * a.pre_bstride + (long)c * a.pre_rstride + t];
            s += u;
        }
        v[i] = u;
    }
    red[grp][lane] = s;
    __syncthreads();
    float tot = red[0][lane];
#pragma unroll
    for (int g2 = 1; g2 < kNormGroups; ++g2) tot += red[g2][lane];
    const float mean = tot / (float)a.c;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = grp + i * kNormGroups;
        if (tv && c < a.c) {
            const float d = v[i] - mean;
            q += d * d;
        }
    }
    red[grp][lane] = q;
    __syncthreads();
    float tot2 = red[0][lane];
#pragma unroll
    for (int g2 = 1; g2 < kNormGroups; ++g2) tot2 += red[g2][lane];
    const float var = tot2 / (float)a.c;
    const float rstd = 1.0f / sqrtf(var + a.eps);
    const float om = (a.out_mask && tv) ? a.out_mask[(long)b * a.t + t] : 1.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = grp + i * kNormGroups;
        if (tv && c < a.c) {
            float o = (v[i] - mean) * rstd * a.gamma[c] + a.beta[c];
            if (a.act == TTSAMD_ACT_RELU) o = fmaxf(o, 0.f);
            else if (a.act == TTSAMD_ACT_GELU) o = o * 0.5f * (1.0f + erff(o * 0.70710678118654752440f));
            if (a.post_res) o = a.post_res[(long)b * a.post_bstride + (long)c * a.post_rstride + t] + o;
            if (a.out_mask) o *= om;
            a.y[(long)b * a.y_bstride + (long)c * a.y_rstride + t] = o;
        }
    }
}

// Text-length tensors (T of a few hundred columns: every norm of the text encoder / duration predictors): the launch is a
// handful of blocks and its time is one block's dependent chain, so the tile is 16 time columns x 64 channel groups — a
// thread owns C/64 channels (3 at C = 192) instead of C/16, four times the blocks — and the depthwise prologue walks the
// taps in the OUTER loop (one column index, validity and mask value per tap instead of one per tap and channel).
// A wavefront = 16 time lanes x 4 channel groups: group sums first meet across the wave (two shuffles), then across the 16
// waves through LDS in a fixed order (deterministic).
template <int NC>   // channels per thread (C <= 64 * NC)
__global__ __launch_bounds__(1024) void channel_norm_small_kernel(const ttsamd_norm_args a)
{
    __shared__ float red[2][16][16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int tl = lane & 15;
    const int grp = wave * 4 + (lane >> 4);     // channel group: channels grp, grp + 64, ...
    const int b = blockIdx.y;
    const int t = blockIdx.x * 16 + tl;
    const bool tv = t < a.t;
    const float *xb = a.x + (long)b * a.x_bstride;
    const float *im = a.in_mask ? a.in_mask + (long)b * a.t : nullptr;

    // Every operand through a buffer resource, validity in the OFFSET (an invalid lane reads 0 from the range check): written as
    // `if (valid) v = p[i]` / `valid ? p[i] : 0` hipcc splits the wave into two paths per element with a full s_waitcnt vmcnt(0)
    // between consecutive loads — this launch was a chain of up to twelve memory round trips (round 6: 5-8 us -> the launch floor)
    constexpr int kInv = (int)0x80000000u;
    const long xslab = ((long)(a.c - 1) * a.x_rstride + a.t) * 4;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xb, xslab);
    float v[NC];
    bool cv[NC];
    int cofs[NC];          // channel index * 4, or out of range
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        cv[i] = tv && (grp + i * 64 < a.c);
        cofs[i] = cv[i] ? (grp + i * 64) * 4 : kInv;
    }
    if (a.dw_w) {
        const int half = (a.dw_kernel - 1) / 2;
        const __amdgpu_buffer_rsrc_t rdb = make_rsrc(a.dw_bias, a.dw_bias ? (long)a.c * 4 : 0);
        const __amdgpu_buffer_rsrc_t rdw = make_rsrc(a.dw_w, (long)a.c * a.dw_kernel * 4);
        const __amdgpu_buffer_rsrc_t rim = make_rsrc(im, im ? (long)a.t * 4 : 0);
#pragma unroll
        for (int i = 0; i < NC; ++i) v[i] = ld_buf(rdb, cofs[i], 0);
        auto tap = [&](int k) {
            const int tt = t + (k - half) * a.dw_dilation;
            const bool ok = tt >= 0 && tt < a.t;
            const float mk = im ? ld_buf(rim, ok ? tt * 4 : kInv, 0) : 1.f;
            float xv[NC], wv[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                xv[i] = ld_buf(rx, (cv[i] && ok) ? ((grp + i * 64) * (int)a.x_rstride + tt) * 4 : kInv, 0);
                wv[i] = ld_buf(rdw, (cv[i] && ok) ? ((grp + i * 64) * a.dw_kernel + k) * 4 : kInv, 0);
            }
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                if (im) xv[i] *= mk;
                // (a tap outside the tensor contributed nothing before; it contributes w = 0 times x = 0 now: same value)
                v[i] = (cv[i] && ok) ? v[i] + wv[i] * xv[i] : v[i];
            }
        };
        if (a.dw_kernel == 3) {      // DDSConv's kernel size (stochastic_duration_predictor.py:46-63): all nine loads in flight together
            tap(0);
            tap(1);
            tap(2);
        } else {
            for (int k = 0; k < a.dw_kernel; ++k) tap(k);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NC; ++i) v[i] = ld_buf(rx, cv[i] ? ((grp + i * 64) * (int)a.x_rstride + t) * 4 : kInv, 0);
    }
    if (a.pre_res) {
        const __amdgpu_buffer_rsrc_t rp = make_rsrc(a.pre_res + (long)b * a.pre_bstride, ((long)(a.c - 1) * a.pre_rstride + a.t) * 4);
        float pr[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) pr[i] = ld_buf(rp, cv[i] ? ((grp + i * 64) * (int)a.pre_rstride + t) * 4 : kInv, 0);
#pragma unroll
        for (int i = 0; i < NC; ++i) v[i] += pr[i];
    }
    // epilogue operands requested before the statistics passes (their latency hides under the two reductions)
    float gam[NC], bet[NC], pres[NC];
    {
        const __amdgpu_buffer_rsrc_t rg = make_rsrc(a.gamma, (long)a.c * 4), rbt = make_rsrc(a.beta, (long)a.c * 4);
        const __amdgpu_buffer_rsrc_t rpo = make_rsrc(a.post_res ? a.post_res + (long)b * a.post_bstride : nullptr,
                                                     a.post_res ? ((long)(a.c - 1) * a.post_rstride + a.t) * 4 : 0);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            gam[i] = ld_buf(rg, cofs[i], 0);
            bet[i] = ld_buf(rbt, cofs[i], 0);
            pres[i] = ld_buf(rpo, cv[i] ? ((grp + i * 64) * (int)a.post_rstride + t) * 4 : kInv, 0);
        }
    }
    const __amdgpu_buffer_rsrc_t rom = make_rsrc(a.out_mask ? a.out_mask + (long)b * a.t : nullptr, a.out_mask ? (long)a.t * 4 : 0);
    const float om = a.out_mask ? ld_buf(rom, tv ? t * 4 : kInv, 0) : 1.f;

    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) s += cv[i] ? v[i] : 0.f;
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (lane < 16) red[0][wave][tl] = s;
    __syncthreads();
    float tot = red[0][0][tl];
#pragma unroll
    for (int w = 1; w < 16; ++w) tot += red[0][w][tl];
    const float mean = tot / (float)a.c;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const float d = v[i] - mean;
        q += cv[i] ? d * d : 0.f;
    }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    if (lane < 16) red[1][wave][tl] = q;
    __syncthreads();
    float tot2 = red[1][0][tl];
#pragma unroll
    for (int w = 1; w < 16; ++w) tot2 += red[1][w][tl];
    const float rstd = 1.0f / sqrtf(tot2 / (float)a.c + a.eps);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        if (cv[i]) {
            float o = (v[i] - mean) * rstd * gam[i] + bet[i];
            if (a.act == TTSAMD_ACT_RELU) o = fmaxf(o, 0.f);
            else if (a.act == TTSAMD_ACT_GELU) o = o * 0.5f * (1.0f + erff(o * 0.70710678118654752440f));
            if (a.post_res) o = pres[i] + o;
            if (a.out_mask) o *= om;
            a.y[(long)b * a.y_bstride + (long)(grp + i * 64) * a.y_rstride + t] = o;
        }
    }
}

// Any channel count (C > 512: no thread can keep its channels in registers): the same 64-column x 16-group tile, but the
// value of a (channel, column) — input, depthwise prologue, pre-residual — is re-evaluated in each of the three passes
// (mean, variance, apply).  Same formulas and the same fixed-order reductions as channel_norm_kernel; a fallback, not a
// fast path (normalization.py:5-53 has no bound on `channels`).
__global__ __launch_bounds__(64 * 16) void channel_norm_any_kernel(const ttsamd_norm_args a)
{
    constexpr int kG = 16;
    __shared__ float red[kG][64];
    const int lane = threadIdx.x;
    const int grp = threadIdx.y;
    const int b = blockIdx.y;
    const int t = blockIdx.x * 64 + lane;
    const bool tv = t < a.t;
    const float *xb = a.x + (long)b * a.x_bstride;
    const float *im = a.in_mask ? a.in_mask + (long)b * a.t : nullptr;
    auto value = [&](int c) {
        float u;
        if (a.dw_w) {
            u = a.dw_bias ? a.dw_bias[c] : 0.f;
            const int half = (a.dw_kernel - 1) / 2;
            for (int k = 0; k < a.dw_kernel; ++k) {
                const int tt = t + (k - half) * a.dw_dilation;
                if (tt >= 0 && tt < a.t) {
                    float xv = xb[(long)c * a.x_rstride + tt];
                    if (im) xv *= im[tt];
                    u += a.dw_w[c * a.dw_kernel + k] * xv;
                }
            }
        } else {
            u = xb[(long)c * a.x_rstride + t];
        }
        if (a.pre_res) u += a.pre_res[(long)b * a.pre_bstride + (long)c * a.pre_rstride + t];
        return u;
    };
    float s = 0.f;
    if (tv)
        for (int c = grp; c < a.c; c += kG) s += value(c);
    red[grp][lane] = s;
    __syncthreads();
    float tot = red[0][lane];
    for (int g2 = 1; g2 < kG; ++g2) tot += red[g2][lane];
    const float mean = tot / (float)a.c;
    __syncthreads();
    float q = 0.f;
    if (tv)
        for (int c = grp; c < a.c; c += kG) {
            const float d = value(c) - mean;
            q += d * d;
        }
    red[grp][lane] = q;
    __syncthreads();
    float tot2 = red[0][lane];
    for (int g2 = 1; g2 < kG; ++g2) tot2 += red[g2][lane];
    const float rstd = 1.0f / sqrtf(tot2 / (float)a.c + a.eps);
    const float om = (a.out_mask && tv) ? a.out_mask[(long)b * a.t + t] : 1.f;
    if (tv)
        for (int c = grp; c < a.c; c += kG) {
            float o = (value(c) - mean) * rstd * a.gamma[c] + a.beta[c];
            if (a.act == TTSAMD_ACT_RELU) o = fmaxf(o, 0.f);
            else if (a.act == TTSAMD_ACT_GELU) o = o * 0.5f * (1.0f + erff(o * 0.70710678118654752440f));
            if (a.post_res) o = a.post_res[(long)b * a.post_bstride + (long)c * a.post_rstride + t] + o;
            if (a.out_mask) o *= om;
            a.y[(long)b * a.y_bstride + (long)c * a.y_rstride + t] = o;
        }
}

}  // namespace ttsamd
using namespace ttsamd;

extern "C" int ttsamd_channel_norm(const ttsamd_norm_args *args, void *stream)
{
    TTSAMD_CHECK_ARG(args, "channel_norm: NULL args");
    const ttsamd_norm_args &a = *args;
    TTSAMD_CHECK_ARG(a.x && a.y && a.gamma && a.beta, "channel_norm: NULL tensor");
    TTSAMD_CHECK_ARG(a.c > 0 && a.t >= 0 && a.batch >= 0, "channel_norm: bad shape");
    TTSAMD_CHECK_ARG(!a.dw_w || (a.dw_kernel > 0 && (a.dw_kernel & 1) && a.dw_dilation > 0),
                     "channel_norm: depthwise prologue needs an odd kernel and dilation > 0");
    TTSAMD_CHECK_ARG(a.act == TTSAMD_ACT_NONE || a.act == TTSAMD_ACT_RELU || a.act == TTSAMD_ACT_GELU,
                     "channel_norm: bad act %d", a.act);
    if (a.batch == 0 || a.t == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(a.batch <= 65535, "channel_norm: batch > 65535");
    hipStream_t st = as_stream(stream);
    if (a.c > 512) {   // y must not alias x here: the passes re-read x (the register-resident kernels read it once)
        TTSAMD_CHECK_ARG(a.y != a.x, "channel_norm: C > 512 cannot run in place");
        hipLaunchKernelGGL(channel_norm_any_kernel, dim3((a.t + 63) / 64, a.batch), dim3(64, 16), 0, st, a);
        TTSAMD_LAUNCH_CHECK();
        return TTSAMD_OK;
    }
    // text-length tensors: 16-column tiles (chosen by T alone, never by the batch: row b of a batch stays bitwise the B = 1 run)
    // (its operands go through 32-bit buffer offsets: per-item slabs below 2 GiB — always, short of a pathological row stride)
    auto slab_ok = [&](long rstride) { return ((long)(a.c - 1) * rstride + a.t) * 4 < 0x7FFFFFF0l; };
    if (a.t <= kNormSmallT && slab_ok(a.x_rstride) && (!a.pre_res || slab_ok(a.pre_rstride)) && (!a.post_res || slab_ok(a.post_rstride))) {
        const dim3 sgrid((a.t + 15) / 16, a.batch);
        if (a.c <= 192) hipLaunchKernelGGL((channel_norm_small_kernel<3>), sgrid, dim3(1024), 0, st, a);
        else if (a.c <= 256) hipLaunchKernelGGL((channel_norm_small_kernel<4>), sgrid, dim3(1024), 0, st, a);
        else hipLaunchKernelGGL((channel_norm_small_kernel<8>), sgrid, dim3(1024), 0, st, a);
        TTSAMD_LAUNCH_CHECK();
        return TTSAMD_OK;
    }
    const dim3 grid((a.t + 63) / 64, a.batch);
    // 16 channel groups x 64 time lanes = 1024 threads per block: the launch is latency-bound (a few MB), so the
    // per-thread dependent chain is kept short (12..32 channels) rather than the block count high
    if (a.c <= 192) hipLaunchKernelGGL((channel_norm_kernel<12, 16>), grid, dim3(64, 16), 0, st, a);
    else if (a.c <= 256) hipLaunchKernelGGL((channel_norm_kernel<16, 16>), grid, dim3(64, 16), 0, st, a);
    else hipLaunchKernelGGL((channel_norm_kernel<32, 16>), grid, dim3(64, 16), 0, st, a);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}
