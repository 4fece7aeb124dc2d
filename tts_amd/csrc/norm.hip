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

#include <cstdlib>

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

    float v[NC];
    bool cv[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        cv[i] = tv && (grp + i * 64 < a.c);
        v[i] = 0.f;
    }
    if (a.dw_w) {
        const int half = (a.dw_kernel - 1) / 2;
#pragma unroll
        for (int i = 0; i < NC; ++i)
            if (cv[i] && a.dw_bias) v[i] = a.dw_bias[grp + i * 64];
        for (int k = 0; k < a.dw_kernel; ++k) {
            const int tt = t + (k - half) * a.dw_dilation;
            const bool ok = tt >= 0 && tt < a.t;
            const float mk = (ok && im) ? im[tt] : 1.f;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                if (cv[i] && ok) {
                    const int c = grp + i * 64;
                    float xv = xb[(long)c * a.x_rstride + tt];
                    if (im) xv *= mk;
                    v[i] += a.dw_w[c * a.dw_kernel + k] * xv;
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < NC; ++i)
            if (cv[i]) v[i] = xb[(long)(grp + i * 64) * a.x_rstride + t];
    }
    if (a.pre_res) {
#pragma unroll
        for (int i = 0; i < NC; ++i)
            if (cv[i]) v[i] += a.pre_res[(long)b * a.pre_bstride + (long)(grp + i * 64) * a.pre_rstride + t];
    }
    // epilogue operands requested before the statistics passes (their latency hides under the two reductions)
    float gam[NC], bet[NC], pres[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = grp + i * 64;
        gam[i] = cv[i] ? a.gamma[c] : 0.f;
        bet[i] = cv[i] ? a.beta[c] : 0.f;
        pres[i] = (cv[i] && a.post_res) ? a.post_res[(long)b * a.post_bstride + (long)c * a.post_rstride + t] : 0.f;
    }
    const float om = (a.out_mask && tv) ? a.out_mask[(long)b * a.t + t] : 1.f;

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


// ---- [norm ->] 1x1 conv -> norm in ONE launch (round 6) --------------------------------------------------------------------
// A DilatedDepthSeparableConv layer (stochastic_duration_predictor.py:46-63) is  x + gelu(LN2(conv1x1(gelu(LN1(dwconv(x*mask))))))
// — three launches of 17 blocks on text-length tensors, ~20 us of which ~15 are launch / drain / memory round trips; the text
// encoder's  LN(x + conv_o(att))  (transformer.py:419-423) is two.  Here one block of 16 columns does the lot:
//   stage 1 (optional) = channel_norm_small_kernel's body (depthwise prologue, LN1, activation), written to LDS instead of HBM as
//            U4[c / 4][column][c % 4];
//   stage 2 = the 1x1 conv on the fp32-input MFMA v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation: an FMA
//            chain per output): wave w owns output rows 16 w .. 16 w + 15; the contraction index is walked as (group of 16, lane quad
//            q, e) -> 16 g + 4 q + e, so a lane's A operand of a group is ONE 16-byte global load of the row-major weight and its B
//            operand ONE ds_read_b128;
//   stage 3 = LN2 / activation / residuals / mask on the accumulator layout (row 16 w + 4 q + r, column = lane % 16), the same
//            fixed-order reductions (quad shuffles, then waves through LDS).
// C % 16 == 0, C <= 256 (16 waves), T <= kNormSmallT; chosen by (C, T) alone, so a batch row stays bitwise its B = 1 run.
using f32x4n = __attribute__((ext_vector_type(4))) float;
#ifdef TTSAMD_PW_CLOCKS   // phase clocks of the fused launch (debug builds only: scripts/pw_norm_clocks.py)
__device__ unsigned long long g_pw_clk[64][16][8];
#define PW_CLK(i)                                                                                         \
    do {                                                                                                  \
        if (lane == 0 && blockIdx.x < 64 && blockIdx.y == 0) g_pw_clk[blockIdx.x][wave][i] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define PW_CLK(i)
#endif

template <int NC>   // channels per thread in stage 1 (C <= 64 * NC)
__global__ __launch_bounds__(1024) void pw_norm_small_kernel(const ttsamd_pw_norm_args a)
{
    __shared__ float red[2][16][16];
    extern __shared__ __attribute__((aligned(16))) float u4[];      // [C / 4][16][4]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int tl = lane & 15;
    const int q = lane >> 4;
    const int grp = wave * 4 + q;               // stage 1: channels grp, grp + 64, ...
    const int b = blockIdx.y;
    const int t = blockIdx.x * 16 + tl;
    const bool tv = t < a.t;
    const int C = a.c;
    constexpr int kOob = kBufOob;
    PW_CLK(0);
    // Every global access goes through a buffer resource with the invalid lanes at an out-of-range offset (reads 0, stores dropped): the
    // whole request list of the block is straight-line code, in flight together.  (As per-lane branches around plain loads — the first
    // version — hipcc waited for each load inside its branch: ~20 serialised round trips, 12 us for a launch that computes for 2.)
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)b * a.x_bstride, ((long)(C - 1) * a.x_rstride + a.t) * 4);
    const __amdgpu_buffer_rsrc_t rim = make_rsrc(a.in_mask ? a.in_mask + (long)b * a.t : nullptr, a.in_mask ? (long)a.t * 4 : 0);
    const __amdgpu_buffer_rsrc_t rom = make_rsrc(a.out_mask ? a.out_mask + (long)b * a.t : nullptr, a.out_mask ? (long)a.t * 4 : 0);
    const __amdgpu_buffer_rsrc_t rpre = make_rsrc(a.pre_res ? a.pre_res + (long)b * a.pre_bstride : nullptr, a.pre_res ? ((long)(C - 1) * a.pre_rstride + a.t) * 4 : 0);
    const __amdgpu_buffer_rsrc_t rpost = make_rsrc(a.post_res ? a.post_res + (long)b * a.post_bstride : nullptr, a.post_res ? ((long)(C - 1) * a.post_rstride + a.t) * 4 : 0);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.pw_w, (long)C * C * 4);
    const __amdgpu_buffer_rsrc_t rpb = make_rsrc(a.pw_b, a.pw_b ? C * 4 : 0);
    const __amdgpu_buffer_rsrc_t rg2 = make_rsrc(a.gamma2, C * 4), rb2 = make_rsrc(a.beta2, C * 4);
    const bool first = a.has_first != 0;
    const bool dw = first && a.dw_w != nullptr;
    const __amdgpu_buffer_rsrc_t rg1 = make_rsrc(first ? a.gamma1 : nullptr, first ? C * 4 : 0), rb1 = make_rsrc(first ? a.beta1 : nullptr, first ? C * 4 : 0);
    const __amdgpu_buffer_rsrc_t rdw = make_rsrc(dw ? a.dw_w : nullptr, dw ? (long)C * a.dw_kernel * 4 : 0);
    const __amdgpu_buffer_rsrc_t rdb = make_rsrc((dw && a.dw_bias) ? a.dw_bias : nullptr, (dw && a.dw_bias) ? C * 4 : 0);

    // ---- requests: the weight fragments of this wave's row tile and the stage-3 operands first (needed last), then stage 1's input ----
    bool cv[NC];
    int ch[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        ch[i] = grp + i * 64;
        cv[i] = tv && ch[i] < C;
    }
    const int nmt = C >> 4;                     // 16-row output tiles = active MFMA waves = groups of 16 along the contraction
    const bool mw = wave < nmt;
    const int row_a = wave * 16 + tl;           // A operand: this lane's weight row
    f32x4n aw[16];
#pragma unroll
    for (int g = 0; g < 16; ++g)
        aw[g] = __builtin_bit_cast(f32x4n, __builtin_amdgcn_raw_buffer_load_b128(rw, (mw && g < nmt) ? (row_a * C + 16 * g + 4 * q) * 4 : kOob, 0, 0));
    float pwb[4], gam2[4], bet2[4], pre[4], post[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = wave * 16 + 4 * q + r;
        const bool ok = mw && tv;
        pwb[r] = ld_buf(rpb, mw ? c * 4 : kOob, 0);
        gam2[r] = ld_buf(rg2, mw ? c * 4 : kOob, 0);
        bet2[r] = ld_buf(rb2, mw ? c * 4 : kOob, 0);
        pre[r] = ld_buf(rpre, ok ? (int)(((long)c * a.pre_rstride + t) * 4) : kOob, 0);
        post[r] = ld_buf(rpost, ok ? (int)(((long)c * a.post_rstride + t) * 4) : kOob, 0);
    }
    const float om = a.out_mask ? ld_buf(rom, tv ? t * 4 : kOob, 0) : 1.f;
    float gam[NC], bet[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        gam[i] = ld_buf(rg1, cv[i] ? ch[i] * 4 : kOob, 0);
        bet[i] = ld_buf(rb1, cv[i] ? ch[i] * 4 : kOob, 0);
    }

    float v[NC];
    if (dw) {
        const int K = a.dw_kernel, half = (K - 1) / 2;
#pragma unroll
        for (int i = 0; i < NC; ++i) v[i] = ld_buf(rdb, cv[i] ? ch[i] * 4 : kOob, 0);
        auto tap = [&](int k) {
            const int tt = t + (k - half) * a.dw_dilation;
            const bool ok = tt >= 0 && tt < a.t;
            const float mk = a.in_mask ? ld_buf(rim, ok ? tt * 4 : kOob, 0) : 1.f;
            float xv[NC], wv[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                xv[i] = ld_buf(rx, (cv[i] && ok) ? (int)(((long)ch[i] * a.x_rstride + tt) * 4) : kOob, 0);
                wv[i] = ld_buf(rdw, cv[i] ? (ch[i] * K + k) * 4 : kOob, 0);
            }
#pragma unroll
            for (int i = 0; i < NC; ++i) v[i] += wv[i] * (xv[i] * mk);
        };
        if (K == 3) {           // the reference's kernel size: three taps' requests back to back
            tap(0), tap(1), tap(2);
        } else {
            for (int k = 0; k < K; ++k) tap(k);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NC; ++i) v[i] = ld_buf(rx, cv[i] ? (int)(((long)ch[i] * a.x_rstride + t) * 4) : kOob, 0);
    }
    PW_CLK(1);
#ifdef TTSAMD_PW_CLOCKS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PW_CLK(2);
#endif
    // ---- stage 1: LN1 + activation (channel_norm_small_kernel's reductions) --------------------------------------------------------
    if (first) {
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
        const float mean = tot / (float)C;
        float qq = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const float d = v[i] - mean;
            qq += cv[i] ? d * d : 0.f;
        }
        qq += __shfl_xor(qq, 16);
        qq += __shfl_xor(qq, 32);
        if (lane < 16) red[1][wave][tl] = qq;
        __syncthreads();
        float tot2 = red[1][0][tl];
#pragma unroll
        for (int w = 1; w < 16; ++w) tot2 += red[1][w][tl];
        const float rstd = 1.0f / sqrtf(tot2 / (float)C + a.eps1);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            float o = (v[i] - mean) * rstd * gam[i] + bet[i];
            if (a.act1 == TTSAMD_ACT_RELU) o = fmaxf(o, 0.f);
            else if (a.act1 == TTSAMD_ACT_GELU) o = o * 0.5f * (1.0f + erff(o * 0.70710678118654752440f));
            v[i] = o;
        }
    }
#pragma unroll
    for (int i = 0; i < NC; ++i)
        if (ch[i] < C) u4[((ch[i] >> 2) * 16 + tl) * 4 + (ch[i] & 3)] = cv[i] ? v[i] : 0.f;      // columns beyond T: zeros
    PW_CLK(3);
    __syncthreads();                            // the tile is complete (and the statistics slots are free again)
    PW_CLK(4);

    // ---- stage 2: w[row][col] = sum_k W[row][k] u[k][col] ---------------------------------------------------------------------
    f32x4n acc = {0.f, 0.f, 0.f, 0.f};
    if (mw) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (g < nmt) {
                const f32x4n bu = *reinterpret_cast<const f32x4n *>(u4 + ((g * 4 + q) * 16 + tl) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[g][e], bu[e], acc, 0, 0, 0);
            }
        }
    }
    PW_CLK(5);
    // ---- stage 3 ------------------------------------------------------------------------------------------------------------
    const bool live = mw && tv;
    float w[4];
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        w[r] = (acc[r] + pwb[r]) + pre[r];
        s += live ? w[r] : 0.f;
    }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (lane < 16) red[0][wave][tl] = s;
    __syncthreads();
    float tot = red[0][0][tl];
#pragma unroll
    for (int ww = 1; ww < 16; ++ww) tot += red[0][ww][tl];
    const float mean = tot / (float)C;
    float qq = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float d = w[r] - mean;
        qq += live ? d * d : 0.f;
    }
    qq += __shfl_xor(qq, 16);
    qq += __shfl_xor(qq, 32);
    if (lane < 16) red[1][wave][tl] = qq;
    __syncthreads();
    float tot2 = red[1][0][tl];
#pragma unroll
    for (int ww = 1; ww < 16; ++ww) tot2 += red[1][ww][tl];
    const float rstd = 1.0f / sqrtf(tot2 / (float)C + a.eps2);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.y + (long)b * a.y_bstride, ((long)(C - 1) * a.y_rstride + a.t) * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = wave * 16 + 4 * q + r;
        float o = (w[r] - mean) * rstd * gam2[r] + bet2[r];
        if (a.act2 == TTSAMD_ACT_RELU) o = fmaxf(o, 0.f);
        else if (a.act2 == TTSAMD_ACT_GELU) o = o * 0.5f * (1.0f + erff(o * 0.70710678118654752440f));
        if (a.post_res) o = post[r] + o;
        if (a.out_mask) o *= om;
        st_buf(ry, o, live ? (int)(((long)c * a.y_rstride + t) * 4) : kOob, 0);
    }
    PW_CLK(6);
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
    if (a.t <= kNormSmallT) {
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

static bool pw_norm_shape_ok(int c, int t) { return c > 0 && (c & 15) == 0 && c <= 256 && t > 0 && t <= kNormSmallT; }

// what every host asks before choosing the fused launch; TTSAMD_PW_NORM=0 answers "no" for all shapes (same-box A/B against the
// three-launch form — Python host and C handles follow the same answer, so they stay bitwise equal either way)
extern "C" int ttsamd_pw_norm_supported(int c, int t)
{
    static const bool off = getenv("TTSAMD_PW_NORM") && atoi(getenv("TTSAMD_PW_NORM")) == 0;
    return (!off && pw_norm_shape_ok(c, t)) ? 1 : 0;
}

extern "C" int ttsamd_pw_norm(const ttsamd_pw_norm_args *args, void *stream)
{
    TTSAMD_CHECK_ARG(args, "pw_norm: NULL args");
    const ttsamd_pw_norm_args &a = *args;
    TTSAMD_CHECK_ARG(a.x && a.y && a.pw_w && a.gamma2 && a.beta2, "pw_norm: NULL tensor");
    TTSAMD_CHECK_ARG(!a.has_first || (a.gamma1 && a.beta1), "pw_norm: the first norm needs gamma / beta");
    TTSAMD_CHECK_ARG(a.c > 0 && a.t >= 0 && a.batch >= 0, "pw_norm: bad shape");
    if (a.batch == 0 || a.t == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(pw_norm_shape_ok(a.c, a.t), "pw_norm: c = %d, t = %d outside the fused kernel (c %% 16 == 0, c <= 256, t <= %d)", a.c, a.t,
                     kNormSmallT);
    TTSAMD_CHECK_ARG(!(a.has_first && a.dw_w) || (a.dw_kernel > 0 && (a.dw_kernel & 1) && a.dw_dilation > 0),
                     "pw_norm: depthwise prologue needs an odd kernel and dilation > 0");
    auto okact = [](int v) { return v == TTSAMD_ACT_NONE || v == TTSAMD_ACT_RELU || v == TTSAMD_ACT_GELU; };
    TTSAMD_CHECK_ARG(okact(a.act2) && (!a.has_first || okact(a.act1)), "pw_norm: bad act");
    TTSAMD_CHECK_ARG(a.batch <= 65535, "pw_norm: batch > 65535");
    TTSAMD_CHECK_ARG((reinterpret_cast<uintptr_t>(a.pw_w) & 15) == 0, "pw_norm: pw_w must be 16-byte aligned");
    auto slab_ok = [&](int64_t rstride) { return rstride >= a.t && ((int64_t)(a.c - 1) * rstride + a.t) * 4 < 0x7FFFFFF0ll; };
    TTSAMD_CHECK_ARG(slab_ok(a.x_rstride) && slab_ok(a.y_rstride) && (!a.pre_res || slab_ok(a.pre_rstride)) && (!a.post_res || slab_ok(a.post_rstride)),
                     "pw_norm: a batch item's slab must stay below 2 GiB (32-bit buffer offsets)");
    TTSAMD_CHECK_ARG(a.y != a.x || !(a.has_first && a.dw_w), "pw_norm: with the depthwise prologue y must not alias x (neighbour blocks read x's halo)");
    hipStream_t st = as_stream(stream);
    const dim3 grid((a.t + 15) / 16, a.batch);
    const size_t lds = (size_t)a.c * 16 * sizeof(float);
    if (a.c <= 192) hipLaunchKernelGGL((pw_norm_small_kernel<3>), grid, dim3(1024), lds, st, a);
    else hipLaunchKernelGGL((pw_norm_small_kernel<4>), grid, dim3(1024), lds, st, a);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

#ifdef TTSAMD_PW_CLOCKS
extern "C" int ttsamd_pw_norm_clocks(unsigned long long *out /* host, 64 * 16 * 8 */)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ttsamd::g_pw_clk), sizeof(unsigned long long) * 64 * 16 * 8) == hipSuccess ? TTSAMD_OK : TTSAMD_ERR_HIP;
}
#endif
