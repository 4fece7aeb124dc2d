// Conv1d dispatcher + host-side weight packing (see conv_kernel.h for the kernel design).
#include <cstdlib>
#include "conv_dispatch.h"

#include <cstring>

using namespace ttsamd;

namespace ttsamd {
// small-grid policy (see ttsamd_conv1d_set_small_grid); TTSAMD_CONV_SMALL_GRID=<0..4> overrides the default for A/B runs
int g_conv_small_grid = [] {
    const char *e = getenv("TTSAMD_CONV_SMALL_GRID");
    const int m = e ? atoi(e) : 4;
    return m < 0 ? 0 : (m > 4 ? 4 : m);
}();
// TTSAMD_SMALL_GRID_BLOCKS=<n>: launches of up to n 128x128-class blocks take the small-grid tiles (A/B runs)
// TTSAMD_H2_MID_MIN=<n>: NORMAL convs of n .. small-grid-limit 128x128-class blocks take the three-product kernel on 128 x 64 tiles
long g_conv_h2_mid_min = [] {
    const char *e = getenv("TTSAMD_H2_MID_MIN");
    const long m = e ? atol(e) : 48;
    return m < 0 ? 0 : m;
}();
long g_conv_small_grid_blocks = [] {
    const char *e = getenv("TTSAMD_SMALL_GRID_BLOCKS");
    const long m = e ? atol(e) : 128;
    return m < 0 ? 0 : m;
}();
}

extern "C" int ttsamd_conv1d_set_small_grid(int mode)
{
    const int was = g_conv_small_grid;
    g_conv_small_grid = mode < 0 ? 0 : (mode > 4 ? 4 : mode);
    return was;
}

// host-side packing and the pure-host policy queries (ttsamd_conv1d_tuned / _supported / pack_weights*): pack_host.cpp
extern "C" int ttsamd_conv1d_tuned(int kernel, int dilation);

extern "C" int ttsamd_conv1d(const ttsamd_conv1d_args *args, void *stream)
{
    TTSAMD_CHECK_ARG(args, "conv1d: NULL args");
    const ttsamd_conv1d_args &a = *args;
    TTSAMD_CHECK_ARG(a.x && a.w_packed && a.y, "conv1d: NULL tensor");
    TTSAMD_CHECK_ARG(a.c_in > 0 && a.c_out > 0 && a.batch >= 0 && a.t_in >= 0 && a.t_out >= 0, "conv1d: bad shape");
    TTSAMD_CHECK_ARG(a.mode != TTSAMD_CONV_COUPLE || a.res, "conv1d: COUPLE needs res");
    TTSAMD_CHECK_ARG(a.mode != TTSAMD_CONV_SHUFFLE || a.shuffle_u > 0, "conv1d: SHUFFLE needs shuffle_u");
    TTSAMD_CHECK_ARG(a.mode != TTSAMD_CONV_GATE || (a.c_out % 32) == 0, "conv1d: GATE needs c_out %% 32 == 0 (whole 16 + 16 row tiles)");
    TTSAMD_CHECK_ARG((a.mode != TTSAMD_CONV_COUPLE_AFFINE && a.mode != TTSAMD_CONV_COUPLE_AFFINE_FWD && a.mode != TTSAMD_CONV_COUPLE_AFFINE_MIX) ||
                         ((a.c_out % 32) == 0 && a.res && a.split_row > 0),
                     "conv1d: COUPLE_AFFINE needs c_out %% 32 == 0 (whole 16 + 16 row tiles), res and split_row");
    TTSAMD_CHECK_ARG(a.mode != TTSAMD_CONV_RES_SKIP || (a.res && a.y2 && a.split_row > 0 && a.split_row % 32 == 0),
                     "conv1d: RES_SKIP needs res, y2 and split_row %% 32 == 0");
    TTSAMD_CHECK_ARG(a.mode != TTSAMD_CONV_COUPLE_AFFINE_MIX || (a.y2 && a.y == a.res && a.split_row % 2 == 0 && a.kernel == 1),
                     "conv1d: COUPLE_AFFINE_MIX runs in place (y == res), needs the mix parameters in y2 and an even split_row");
    TTSAMD_CHECK_ARG(a.mode >= 0 && a.mode <= TTSAMD_CONV_COUPLE_AFFINE_MIX, "conv1d: unknown mode %d", a.mode);
    TTSAMD_CHECK_ARG(a.in_act != TTSAMD_ACT_LRELU || (a.in_slope >= 0.f && a.in_slope <= 1.f),
                     "conv1d: leaky-ReLU slope %g outside [0, 1] (the kernels evaluate it as max(v, v * slope))", (double)a.in_slope);
    if (a.batch == 0 || a.t_out == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(a.batch <= 65535, "conv1d: batch > 65535");
    {   // the kernel addresses every per-item slab with 32-bit byte offsets (buffer resources)
        const int64_t lim = 0x7FFFFFF0;
        const int64_t rows_y = (a.mode == TTSAMD_CONV_SHUFFLE) ? (a.c_out / (a.shuffle_u > 0 ? a.shuffle_u : 1)) : a.c_out;
        const int64_t cols_y = (a.mode == TTSAMD_CONV_SHUFFLE) ? a.shuffle_t_out : a.t_out;
        if (((int64_t)a.c_in * a.x_rstride + a.t_in) * 4 >= lim || (rows_y * a.y_rstride + cols_y) * 4 >= lim ||
            (a.res && ((int64_t)a.c_out * a.res_rstride + a.t_out) * 4 >= lim) ||
            (a.accum && ((int64_t)a.c_out * a.accum_rstride + a.t_out) * 4 >= lim)) {
            set_error("conv1d: a per-item [C, T] slab exceeds 2 GiB (time-tile the call)");
            return TTSAMD_ERR_UNSUPPORTED;
        }
    }
    if (!ttsamd_conv1d_supported(a.kernel, a.dilation)) {
        set_error("conv1d: (kernel=%d, dilation=%d) outside the supported range (kernel <= 31, dilation <= 27)", a.kernel, a.dilation);
        return TTSAMD_ERR_UNSUPPORTED;
    }
    hipStream_t st = as_stream(stream);
    if (!ttsamd_conv1d_tuned(a.kernel, a.dilation)) return conv1d_launch_generic(a, st);      // any other (k, d): conv_generic.hip
    if (conv_post_eligible(a)) return conv_post_launch(a, st);   // C -> 1 (HiFiGAN conv_post): pure HBM streaming
    switch (a.kernel) {
        case 1: return conv1d_launch_k1(a, st);
        case 2: return conv1d_launch_k2(a, st);
        case 3: return conv1d_launch_k3(a, st);
        case 5: return conv1d_launch_k5(a, st);
        case 7: return conv1d_launch_k7(a, st);
        case 11: return conv1d_launch_k11(a, st);
    }
    return TTSAMD_ERR_UNSUPPORTED;
}
