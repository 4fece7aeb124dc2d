// Mode / precision dispatch of the conv kernels (one translation unit per kernel size includes this).
#pragma once
#include "conv_kernel.h"
#include "conv_kernel_x3.h"
#include "conv_kernel_x3s.h"
#include "conv_kernel_x3o.h"
#include "conv_kernel_h2.h"

namespace ttsamd {

int conv1d_launch_generic(const ttsamd_conv1d_args &a, hipStream_t st);   // conv_generic.hip: any (kernel, dilation), NORMAL / GATE / SHUFFLE

// a (kernel, dilation) with tuned NORMAL kernels but no instantiation of this mode (a WaveNet gate conv at k = 7, a polyphase
// ConvTranspose with three taps): the generic kernel
inline int conv1d_mode_unsupported(const ttsamd_conv1d_args &a, hipStream_t st) { return conv1d_launch_generic(a, st); }

// Fused epilogues exist where the models use them: GATE on the WaveNet in_layers (k=3/5, d=1), SHUFFLE on the
// polyphase transposed conv (k=2), COUPLE / RES_SKIP / COUPLE_AFFINE on 1x1 convs.
// a.w_split != NULL selects the split-bf16 kernels (conv_kernel_x3.h), NULL the fp32-input MFMA kernels.
template <int K, int D, int MODE>
int conv1d_launch_prec(const ttsamd_conv1d_args &a, hipStream_t st)
{
    if (a.w_split) return conv1d_x3_launch_tiles<K, D, MODE>(a, st);
    return conv1d_launch_tiles<K, D, MODE>(a, st);
}

template <int K, int D>
int conv1d_launch_kd(const ttsamd_conv1d_args &a, hipStream_t st)
{
    switch (a.mode) {
        case TTSAMD_CONV_NORMAL: return conv1d_launch_prec<K, D, TTSAMD_CONV_NORMAL>(a, st);
        case TTSAMD_CONV_GATE:
            if constexpr ((K == 3 || K == 5) && D == 1) return conv1d_launch_prec<K, D, TTSAMD_CONV_GATE>(a, st);
            break;
        case TTSAMD_CONV_SHUFFLE:
            if constexpr (K == 2) return conv1d_launch_prec<K, D, TTSAMD_CONV_SHUFFLE>(a, st);
            break;
        case TTSAMD_CONV_COUPLE:
            if constexpr (K == 1) return conv1d_launch_prec<K, D, TTSAMD_CONV_COUPLE>(a, st);
            break;
        case TTSAMD_CONV_RES_SKIP:
            if constexpr (K == 1) return conv1d_launch_prec<K, D, TTSAMD_CONV_RES_SKIP>(a, st);
            break;
        case TTSAMD_CONV_COUPLE_AFFINE:
            if constexpr (K == 1) return conv1d_launch_prec<K, D, TTSAMD_CONV_COUPLE_AFFINE>(a, st);
            break;
        case TTSAMD_CONV_COUPLE_AFFINE_FWD:
            if constexpr (K == 1) return conv1d_launch_prec<K, D, TTSAMD_CONV_COUPLE_AFFINE_FWD>(a, st);
            break;
        case TTSAMD_CONV_COUPLE_AFFINE_MIX:
            if constexpr (K == 1) return conv1d_launch_prec<K, D, TTSAMD_CONV_COUPLE_AFFINE_MIX>(a, st);
            break;
    }
    return conv1d_mode_unsupported(a, st);
}

// single-output-channel streaming kernel (conv_post.hip)
bool conv_post_eligible(const ttsamd_conv1d_args &a);
int conv_post_launch(const ttsamd_conv1d_args &a, hipStream_t st);

// one translation unit per kernel size (conv_k*.hip) so hipcc compiles them in parallel
int conv1d_launch_k1(const ttsamd_conv1d_args &a, hipStream_t st);
int conv1d_launch_k2(const ttsamd_conv1d_args &a, hipStream_t st);
int conv1d_launch_k3(const ttsamd_conv1d_args &a, hipStream_t st);
int conv1d_launch_k5(const ttsamd_conv1d_args &a, hipStream_t st);
int conv1d_launch_k7(const ttsamd_conv1d_args &a, hipStream_t st);
int conv1d_launch_k11(const ttsamd_conv1d_args &a, hipStream_t st);

}  // namespace ttsamd
