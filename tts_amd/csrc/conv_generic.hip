// Generic Conv1d fallback: ANY kernel size and dilation (runtime values) on the split-bf16 arithmetic of conv_kernel_x3.h.
//
// The tuned kernels are template instantiations per (kernel size, dilation) — HiFiGAN / VITS defaults: k in {1,2,3,5,7,11},
// d in {1,3,5(,9)}.  The reference takes any `resblock_kernel_sizes` / `resblock_dilation_sizes` / `upsample_kernel_sizes`
// (TTS/vocoder/models/hifigan_generator.py:199-233) and any `kernel_size_*` / `dilation_rate_*` (TTS/tts/models/vits.py:544-600):
// every other (k, d) up to k = 31, d = 27 lands here, and so does a ConvTranspose1d whose kernel is not twice its stride (its
// polyphase form is a Conv1d with ceil(k / stride) taps, any count).  Correct, not fast: ONE wave per 32x32 output tile; per
// 16-channel chunk and tap every lane loads, activates and splits its own B fragment (column, 8 channels) of the 32 shifted
// input columns, fetches the tap's weight fragments and issues the six split products — in the chunk-major, tap-minor, smallest-product-first order of the tuned
// kernels, with the same weight image, accumulator layout and fused epilogue (conv_epilogue): NORMAL, GATE and SHUFFLE modes.
#include "conv_dispatch.h"

namespace ttsamd {

template <int MODE>
__global__ __launch_bounds__(64) void conv1d_x3g_kernel(const ttsamd_conv1d_args a)
{
    const int lane = threadIdx.x;
    const int h = lane >> 5;
    const int j = lane & 31;
    const int b = blockIdx.z;
    const int mb = blockIdx.y;
    const int t0 = blockIdx.x * 32;
    const int K = a.kernel, D = a.dilation;
    const int nchunks = (a.c_in + kConvCK - 1) / kConvCK;
    constexpr int kOob = kConvOob;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)b * a.x_bstride, ((long)(a.c_in - 1) * a.x_rstride + a.t_in) * 4);
    const __amdgpu_buffer_rsrc_t rm = make_rsrc(a.in_mask ? a.in_mask + (long)b * a.t_in : nullptr, a.in_mask ? (long)a.t_in * 4 : 0);
    const int row_bytes = (int)a.x_rstride * 4;
    const u32x4 *const wp = reinterpret_cast<const u32x4 *>(a.w_split) + (long)mb * ((long)nchunks * K * 3 * 64) + lane;

    f32x16 acc[1][1];
    const bool folded = conv_acc_init<MODE, 1, 1, 1, 1>(acc, a, b, mb, t0, 0, 0, h, j);
    for (int c = 0; c < nchunks; ++c) {
        for (int tap = 0; tap < K; ++tap) {
            // stage channels [16c, 16c+16) x columns t0 - pad + tap*D + [0, 32): lane = (half, column), 8 channels each
            const int gt = t0 - a.pad_left + tap * D + j;
            const bool ok = gt >= 0 && gt < a.t_in;
            const int off = ok ? (int)(((long)(c * kConvCK + 8 * h) * a.x_rstride + gt) * 4) : kOob;
            const float m = a.in_mask ? ld_buf(rm, ok ? gt * 4 : kOob, 0) : 1.f;
            float st[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) st[i] = ld_buf(rx, off == kOob ? kOob : off + i * row_bytes, 0);
            u32x4 aw[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) aw[q] = wp[((long)c * K + tap) * (3 * 64) + q * 64];
            unsigned pw[3][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                conv_split3x2(conv_in_act(st[2 * i] * m, a.in_act, a.in_slope), conv_in_act(st[2 * i + 1] * m, a.in_act, a.in_slope),
                              pw[0][i], pw[1][i], pw[2][i]);
            // a lane's B fragment of the 32x32x16 MFMA is (column j, channels 8h .. 8h+7): exactly what it has just staged —
            // no LDS round trip (the tuned kernels stage through LDS to SHARE a tile between taps and waves)
            u32x4 bq[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                bq[q].x = pw[q][0];
                bq[q].y = pw[q][1];
                bq[q].z = pw[q][2];
                bq[q].w = pw[q][3];
            }
            constexpr int pa[6] = {2, 1, 0, 1, 0, 0};   // smallest products first (as conv1d_x3_kernel)
            constexpr int pb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[pa[t]]),
                                                                    __builtin_bit_cast(bf16x8, bq[pb[t]]), acc[0][0], 0, 0, 0);
        }
    }
    conv_epilogue<MODE, 1, 1, 1, 1>(acc, b, mb, t0, 0, 0, h, j, folded);
}

template <int MODE>
static int conv1d_generic_mode(const ttsamd_conv1d_args &a, hipStream_t st)
{
    const int mtiles = (a.c_out + 31) / 32;
    const int nblocks = (a.t_out + 31) / 32;
    TTSAMD_CHECK_ARG(mtiles <= 65535, "conv1d (generic): c_out > 2 M");
    hipLaunchKernelGGL(conv1d_x3g_kernel<MODE>, dim3(nblocks, mtiles, a.batch), dim3(64), 0, st, a);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

int conv1d_launch_generic(const ttsamd_conv1d_args &a, hipStream_t st)
{
    if (!a.w_split) {
        set_error("conv1d: kernel=%d dilation=%d runs on the generic split-bf16 kernel, which needs the w_split image", a.kernel,
                  a.dilation);
        return TTSAMD_ERR_UNSUPPORTED;
    }
    switch (a.mode) {
        case TTSAMD_CONV_NORMAL: return conv1d_generic_mode<TTSAMD_CONV_NORMAL>(a, st);
        case TTSAMD_CONV_GATE: return conv1d_generic_mode<TTSAMD_CONV_GATE>(a, st);
        case TTSAMD_CONV_SHUFFLE: return conv1d_generic_mode<TTSAMD_CONV_SHUFFLE>(a, st);
    }
    set_error("conv1d: mode %d has no generic kernel (kernel=%d dilation=%d): the 1x1 coupling / res-skip modes are kernel 1", a.mode,
              a.kernel, a.dilation);
    return TTSAMD_ERR_UNSUPPORTED;
}

}  // namespace ttsamd
