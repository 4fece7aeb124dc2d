// Conv1d dispatcher + host-side weight packing (see conv_kernel.h for the kernel design).
#include <cstdlib>
#include "conv_dispatch.h"

#include <cmath>
#include <cstring>
#include <vector>

using namespace ttsamd;

namespace ttsamd {
// small-grid policy (see ttsamd_conv1d_set_small_grid); TTSAMD_CONV_SMALL_GRID=<0..4> overrides the default for A/B runs
int g_conv_small_grid = [] {
    const char *e = getenv("TTSAMD_CONV_SMALL_GRID");
    const int m = e ? atoi(e) : 4;
    return m < 0 ? 0 : (m > 4 ? 4 : m);
}();
// TTSAMD_SMALL_GRID_BLOCKS=<n>: launches of up to n 128x128-class blocks take the small-grid tiles (A/B runs)
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

// (kernel, dilation) pairs with tuned template instantiations; everything else up to k = 31, d = 27 takes the generic kernel
static int conv1d_tuned(int kernel, int dilation)
{
    switch (kernel) {
        case 1: case 2: case 5: return dilation == 1;
        case 3: case 7: case 11: return dilation == 1 || dilation == 3 || dilation == 5 || (kernel == 3 && dilation == 9);
        default: return 0;
    }
}

extern "C" int ttsamd_conv1d_tuned(int kernel, int dilation) { return conv1d_tuned(kernel, dilation); }

extern "C" int ttsamd_conv1d_supported(int kernel, int dilation)
{
    return kernel >= 1 && kernel <= 31 && dilation >= 1 && dilation <= 27;
}

extern "C" size_t ttsamd_conv1d_packed_floats(int c_out, int c_in, int kernel)
{
    if (c_out <= 0 || c_in <= 0 || kernel <= 0) return 0;
    const size_t mtiles = (size_t)(c_out + 31) / 32;
    const size_t nchunks = (size_t)(c_in + kConvCK - 1) / kConvCK;
    const size_t gpc = (size_t)(kConvCK / 2) * kernel / 4;
    return mtiles * nchunks * gpc * 256 + 256;  // + one zero group of prefetch slack
}

extern "C" int ttsamd_conv1d_pack_weights(float *dst, const float *w, int c_out, int c_in, int kernel)
{
    TTSAMD_CHECK_ARG(dst && w && c_out > 0 && c_in > 0 && kernel > 0, "conv1d_pack_weights: bad args");
    TTSAMD_CHECK_ARG(((kConvCK / 2) * kernel) % 4 == 0, "conv1d_pack_weights: kernel size %d unsupported", kernel);
    const size_t n = ttsamd_conv1d_packed_floats(c_out, c_in, kernel);
    memset(dst, 0, n * sizeof(float));
    const int mtiles = (c_out + 31) / 32;
    const int nchunks = (c_in + kConvCK - 1) / kConvCK;
    const int gpc = (kConvCK / 2) * kernel / 4;
    const long ksg = (long)nchunks * gpc;
    for (int mt = 0; mt < mtiles; ++mt)
        for (int c = 0; c < nchunks; ++c)
            for (int p = 0; p < kConvCK / 2; ++p)
                for (int tap = 0; tap < kernel; ++tap) {
                    const int ksl = p * kernel + tap;
                    const long g = (long)c * gpc + ksl / 4;
                    const int s = ksl % 4;
                    for (int l = 0; l < 64; ++l) {
                        const int row = mt * 32 + (l & 31);
                        const int ci = c * kConvCK + 2 * p + (l >> 5);
                        if (row < c_out && ci < c_in)
                            dst[((mt * ksg + g) * 64 + l) * 4 + s] = w[((long)row * c_in + ci) * kernel + tap];
                    }
                }
    return TTSAMD_OK;
}

// ---- split-bf16 image ---------------------------------------------------------------------------------------------
static inline uint16_t f32_to_bf16_rne(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t h)
{
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

extern "C" size_t ttsamd_conv1d_packed_split_bytes(int c_out, int c_in, int kernel)
{
    if (c_out <= 0 || c_in <= 0 || kernel <= 0) return 0;
    const size_t mtiles = (size_t)(c_out + 31) / 32;
    const size_t nchunks = (size_t)(c_in + kConvCK - 1) / kConvCK;
    return (mtiles * nchunks * kernel + 2) * 3 * 64 * 16;   // + two zero groups of prefetch slack (the fused ResBlock kernel requests two taps ahead)
}

extern "C" int ttsamd_conv1d_pack_weights_split(void *dst_, const float *w, int c_out, int c_in, int kernel)
{
    TTSAMD_CHECK_ARG(dst_ && w && c_out > 0 && c_in > 0 && kernel > 0, "conv1d_pack_weights_split: bad args");
    uint16_t *dst = static_cast<uint16_t *>(dst_);
    memset(dst, 0, ttsamd_conv1d_packed_split_bytes(c_out, c_in, kernel));
    const int mtiles = (c_out + 31) / 32;
    const int nchunks = (c_in + kConvCK - 1) / kConvCK;
    for (int mt = 0; mt < mtiles; ++mt)
        for (int c = 0; c < nchunks; ++c)
            for (int tap = 0; tap < kernel; ++tap) {
                uint16_t *grp = dst + (((size_t)mt * nchunks + c) * kernel + tap) * (3 * 64 * 8);
                for (int l = 0; l < 64; ++l) {
                    const int row = mt * 32 + (l & 31);
                    if (row >= c_out) continue;
                    for (int i = 0; i < 8; ++i) {
                        const int ci = c * kConvCK + 8 * (l >> 5) + i;
                        if (ci >= c_in) continue;
                        const float v = w[((long)row * c_in + ci) * kernel + tap];
                        const uint16_t p1 = f32_to_bf16_rne(v);
                        const float r1 = v - bf16_to_f32(p1);
                        const uint16_t p2 = f32_to_bf16_rne(r1);
                        const float r2 = r1 - bf16_to_f32(p2);
                        const uint16_t p3 = f32_to_bf16_rne(r2);
                        grp[(0 * 64 + l) * 8 + i] = p1;
                        grp[(1 * 64 + l) * 8 + i] = p2;
                        grp[(2 * 64 + l) * 8 + i] = p3;
                    }
                }
            }
    return TTSAMD_OK;
}

// ---- two-part fp16 image (conv_kernel_h2.h) -----------------------------------------------------------------------------
extern "C" size_t ttsamd_conv1d_packed_h2_bytes(int c_out, int c_in, int kernel)
{
    if (c_out <= 0 || c_in <= 0 || kernel <= 0) return 0;
    const size_t mtiles = (size_t)(c_out + 31) / 32;
    return conv_h2_table_offset(c_out, c_in, kernel) + sizeof(H2RowTable) + mtiles * 32 * 2 * sizeof(float);
}

extern "C" int ttsamd_conv1d_pack_weights_h2(void *dst_, const float *w, int c_out, int c_in, int kernel)
{
    TTSAMD_CHECK_ARG(dst_ && w && c_out > 0 && c_in > 0 && kernel > 0, "conv1d_pack_weights_h2: bad args");
    unsigned char *const base = static_cast<unsigned char *>(dst_);
    memset(base, 0, ttsamd_conv1d_packed_h2_bytes(c_out, c_in, kernel));
    const int mtiles = (c_out + 31) / 32;
    const int nchunks = (c_in + kConvCK - 1) / kConvCK;
    H2RowTable *const hdr = reinterpret_cast<H2RowTable *>(base + conv_h2_table_offset(c_out, c_in, kernel));
    float *const tab = reinterpret_cast<float *>(hdr + 1);
    // row exponents: the row's largest magnitude lands in [2^13, 2^14); an all-zero (or padding) row keeps exponent 0
    std::vector<int> rexp((size_t)mtiles * 32, 0);
    int emax = -1000;
    for (int row = 0; row < mtiles * 32; ++row) {
        float mx = 0.f;
        if (row < c_out)
            for (long i = 0; i < (long)c_in * kernel; ++i) {
                const float v = fabsf(w[(long)row * c_in * kernel + i]);
                if (v > mx && v <= 3.4e38f) mx = v;
            }
        int e = 0;
        if (mx > 0.f) {
            int ex;
            frexpf(mx, &ex);              // mx = f * 2^ex, f in [0.5, 1)  ->  mx in [2^(ex-1), 2^ex)
            e = 14 - ex;
            e = e > 126 ? 126 : (e < -126 ? -126 : e);
        }
        rexp[row] = e;
        tab[2 * row] = ldexpf(1.f, e);
        tab[2 * row + 1] = ldexpf(1.f, -e);
        if (row < c_out && e > emax) emax = e;
    }
    hdr->max_row_exp = emax == -1000 ? 0 : emax;
    _Float16 *const dst = reinterpret_cast<_Float16 *>(base);
    for (int mt = 0; mt < mtiles; ++mt)
        for (int c = 0; c < nchunks; ++c)
            for (int tap = 0; tap < kernel; ++tap) {
                _Float16 *grp = dst + (((size_t)mt * nchunks + c) * kernel + tap) * (2 * 64 * 8);
                for (int l = 0; l < 64; ++l) {
                    const int row = mt * 32 + (l & 31);
                    if (row >= c_out) continue;
                    for (int i = 0; i < 8; ++i) {
                        const int ci = c * kConvCK + 8 * (l >> 5) + i;
                        if (ci >= c_in) continue;
                        const float v = ldexpf(w[((long)row * c_in + ci) * kernel + tap], rexp[row]);     // exact
                        const _Float16 hi = (_Float16)v;                                                 // round to nearest even
                        const _Float16 lo = (_Float16)((v - (float)hi) * 2048.f);                        // residual exact
                        grp[(0 * 64 + l) * 8 + i] = hi;
                        grp[(1 * 64 + l) * 8 + i] = lo;
                    }
                }
            }
    return TTSAMD_OK;
}

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
    if (!conv1d_tuned(a.kernel, a.dilation)) return conv1d_launch_generic(a, st);      // any other (k, d): conv_generic.hip
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
