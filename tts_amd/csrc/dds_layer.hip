// One DilatedDepthSeparableConv layer (TTS/tts/layers/vits/stochastic_duration_predictor.py:46-63) as ONE launch, for the
// text-length tensors of a single request (round 4):
//
//     y = GELU(LN_c(depthwise_k(x * mask)))          convs_sep[i] -> norms_1[i] -> gelu
//     y = GELU(LN_c(W y + b))                        convs_1x1[i] -> norms_2[i] -> gelu
//     x = x + y            [* mask after the last layer]
//
// Before: three launches per layer (channel_norm with the depthwise prologue, the 1x1 conv, channel_norm with the residual), 36
// of the ~95 launches of a VITS request's front end, ~6 us each.  Here a block owns 32 columns and all 192 channels: 12 waves;
// in the first phase wave w owns channels 16w .. 16w+15 (lane = 8-channel half x column: exactly a B fragment of the 32x32x16
// MFMA), in the second wave (m, kh) owns output rows 32m .. 32m+31 and half of the reduction.  Every global request — input
// taps, the 1x1 conv's weight fragments, the residual rows, norm parameters — is issued before the first barrier; after that the
// block only talks to LDS.  LayerNorm statistics are two-pass (mean, then centred squares), partial sums meet in LDS in a fixed
// order.  Arithmetic of the 1x1 conv: the split-bf16 scheme of conv_kernel_x3.h.
#include "conv_kernel_x3.h"

namespace ttsamd {

using u32x2d = __attribute__((ext_vector_type(2))) unsigned;

constexpr int kDdsC = 192;
constexpr int kDdsNCH = kDdsC / 16;      // 12 chunks = 12 waves
constexpr int kDdsMT = kDdsC / 32;       // 6 m-tiles
constexpr int kDdsPlane = 32 * 16;
constexpr int kDdsThreads = 64 * 12;
constexpr int kDdsMaxK = 7;

__device__ __forceinline__ float dds_gelu(float o) { return o * 0.5f * (1.0f + erff(o * 0.70710678118654752440f)); }

__global__ __launch_bounds__(kDdsThreads) void dds_layer_kernel(const ttsamd_dds_layer_args a)
{
    __shared__ __attribute__((aligned(16))) unsigned char planes[3 * kDdsNCH * 2 * kDdsPlane];   // GELU(LN1(dw)) as B operand
    float(*const redt)[16][64] = reinterpret_cast<float(*)[16][64]>(planes);    // partial tiles of K half 1: over the planes, once every
                                                                                // wave is done reading them (24.6 of 36.9 KB)
    __shared__ float red1[24][32], red2[24][32];                                                  // LayerNorm partial sums

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int j = lane & 31;
    const int m = wave % kDdsMT;
    const int kh = wave / kDdsMT;
    const int b = blockIdx.y;
    const int T = a.t;
    const int t = blockIdx.x * 32 + j;
    const bool tv = t < T;
    const float *const xb = a.x + (long)b * kDdsC * T;
    const float *const mrow = a.mask ? a.mask + (long)b * T : nullptr;

    // ---- every global request of the block ------------------------------------------------------------------------------
    // 1x1 conv weights of (m-tile m, chunks kh*6 .. kh*6+5)
    u32x4 aw[6][3];
    {
        const u32x4 *wp = reinterpret_cast<const u32x4 *>(a.w_split) + (long)m * (kDdsNCH * 3 * 64) + lane;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q) aw[i][q] = wp[(long)(kh * 6 + i) * (3 * 64) + q * 64];
    }
    // depthwise taps of channels 16 wave + 8 h + i at column t
    const int K = a.dw_kernel, halfk = (K - 1) / 2;
    float u[8];
    const int c0 = wave * 16 + 8 * h;
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = a.dw_bias ? a.dw_bias[c0 + i] : 0.f;
#pragma unroll
    for (int k = 0; k < kDdsMaxK; ++k) {
        if (k < K) {
            const int tt = t + (k - halfk) * a.dw_dilation;
            const bool ok = tv && tt >= 0 && tt < T;
            const float mv = ok ? (mrow ? mrow[tt] : 1.f) : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xv = ok ? xb[(long)(c0 + i) * T + tt] : 0.f;
                u[i] += a.dw_w[(c0 + i) * K + k] * (xv * mv);
            }
        }
    }
    float g1[8], be1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        g1[i] = a.gamma1[c0 + i];
        be1[i] = a.beta1[c0 + i];
    }
    // second-phase operands of the waves that finish a tile (kh == 0): residual rows, bias, norm parameters
    float xres[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        xres[r] = (kh == 0 && tv) ? xb[(long)row * T + t] : 0.f;
    }
    const float om = (a.out_mask && tv) ? a.out_mask[(long)b * T + t] : 1.f;

    // ---- phase A: LayerNorm over channels of the depthwise output, GELU, split -> LDS planes ---------------------------------
    {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += u[i];
        red1[wave * 2 + h][j] = s;
        __syncthreads();
        float tot = red1[0][j];
#pragma unroll
        for (int g = 1; g < 24; ++g) tot += red1[g][j];
        const float mean = tot / (float)kDdsC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float d = u[i] - mean;
            q += d * d;
        }
        red2[wave * 2 + h][j] = q;
        __syncthreads();
        float tot2 = red2[0][j];
#pragma unroll
        for (int g = 1; g < 24; ++g) tot2 += red2[g][j];
        const float rstd = 1.0f / sqrtf(tot2 / (float)kDdsC + a.eps);
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = dds_gelu((u[i] - mean) * rstd * g1[i] + be1[i]);
        unsigned pw[3][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) conv_split3x2(o[2 * i], o[2 * i + 1], pw[0][i], pw[1][i], pw[2][i]);
#pragma unroll
        for (int q3 = 0; q3 < 3; ++q3) {
            u32x4 w;
            w.x = pw[q3][0];
            w.y = pw[q3][1];
            w.z = pw[q3][2];
            w.w = pw[q3][3];
            *reinterpret_cast<u32x4 *>(planes + ((q3 * kDdsNCH + wave) * 2 + h) * kDdsPlane + j * 16) = w;
        }
    }
    __syncthreads();

    // ---- phase B: 1x1 conv (tile m, K half kh), halves meet in LDS ------------------------------------------------------------
    // (bias and the second norm's parameters: 3 x 192 floats, cache-resident; requested here, they arrive under the MFMAs)
    float bia[16], g2[16], be2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        bia[r] = a.bias ? a.bias[row] : 0.f;
        g2[r] = a.gamma2[row];
        be2[r] = a.beta2[row];
    }
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc[r] = 0.f;
        acc2[r] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        u32x4 bq[3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
            bq[q] = *reinterpret_cast<const u32x4 *>(planes + ((q * kDdsNCH + kh * 6 + i) * 2 + h) * kDdsPlane + j * 16);
        constexpr int pa[6] = {2, 1, 0, 1, 0, 0};
        constexpr int pb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int tq = 0; tq < 6; ++tq) {
            if (tq & 1)
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[i][pa[tq]]),
                                                               __builtin_bit_cast(bf16x8, bq[pb[tq]]), acc2, 0, 0, 0);
            else
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[i][pa[tq]]),
                                                              __builtin_bit_cast(bf16x8, bq[pb[tq]]), acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    __syncthreads();                                         // the planes are dead: their bytes carry the partial tiles now
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) redt[m][r][lane] = acc[r];
    }
    __syncthreads();

    // ---- phase C: + bias, LayerNorm over channels, GELU, + x [, * mask] ------------------------------------------------------
    float v[16];
    float s = 0.f;
    if (kh == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[r] = (acc[r] + redt[m][r][lane]) + bia[r];
            s += v[r];
        }
        red1[m * 2 + h][j] = s;
    }
    __syncthreads();
    float mean = 0.f, q = 0.f;
    if (kh == 0) {
        float tot = red1[0][j];
#pragma unroll
        for (int g = 1; g < 12; ++g) tot += red1[g][j];
        mean = tot / (float)kDdsC;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = v[r] - mean;
            q += d * d;
        }
        red2[m * 2 + h][j] = q;
    }
    __syncthreads();
    if (kh == 0) {
        float tot2 = red2[0][j];
#pragma unroll
        for (int g = 1; g < 12; ++g) tot2 += red2[g][j];
        const float rstd = 1.0f / sqrtf(tot2 / (float)kDdsC + a.eps);
        float *const yb = a.y + (long)b * kDdsC * T;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float o = dds_gelu((v[r] - mean) * rstd * g2[r] + be2[r]);
            o = xres[r] + o;
            if (a.out_mask) o *= om;
            if (tv) yb[(long)row * T + t] = o;
        }
    }
}

}  // namespace ttsamd
using namespace ttsamd;

extern "C" int ttsamd_dds_layer_supported(int channels, int dw_kernel) { return channels == kDdsC && dw_kernel >= 1 && dw_kernel <= kDdsMaxK && (dw_kernel & 1); }

extern "C" int ttsamd_dds_layer(const ttsamd_dds_layer_args *args, void *stream)
{
    TTSAMD_CHECK_ARG(args, "dds_layer: NULL args");
    const ttsamd_dds_layer_args &a = *args;
    TTSAMD_CHECK_ARG(a.x && a.y && a.dw_w && a.gamma1 && a.beta1 && a.w_split && a.gamma2 && a.beta2, "dds_layer: NULL tensor");
    TTSAMD_CHECK_ARG(a.x != a.y, "dds_layer: y must not alias x (neighbouring tiles read x's depthwise halo)");
    TTSAMD_CHECK_ARG(ttsamd_dds_layer_supported(a.c, a.dw_kernel), "dds_layer: built for %d channels and odd depthwise kernels <= %d (got %d, %d)",
                     kDdsC, kDdsMaxK, a.c, a.dw_kernel);
    TTSAMD_CHECK_ARG(a.dw_dilation >= 1 && a.t >= 0 && a.batch >= 0 && a.batch <= 65535, "dds_layer: bad shape");
    if (a.t == 0 || a.batch == 0) return TTSAMD_OK;
    hipLaunchKernelGGL(dds_layer_kernel, dim3((a.t + 31) / 32, a.batch), dim3(kDdsThreads), 0, as_stream(stream), a);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}
