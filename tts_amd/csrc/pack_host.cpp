// Host-side half of the conv / ResBlock ABI that has no kernel behind it: weight packing into MFMA fragment order (fp32,
// split-bf16 and two-part fp16 images) and the policy queries (which shapes are tuned / supported / fusable).  Plain C++ — no
// HIP headers — so that it also builds stand-alone under -fsanitize=address,undefined (tests/test_host_cpu.py).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/tts_amd.h"
#include "pack_layout.h"

namespace ttsamd {
void set_error(const char *fmt, ...);      // common.hip (the sanitizer driver brings its own)
}
using namespace ttsamd;

#define TTSAMD_CHECK_ARG(cond, ...)          \
    do {                                     \
        if (!(cond)) {                       \
            ::ttsamd::set_error(__VA_ARGS__); \
            return TTSAMD_ERR_INVALID;       \
        }                                    \
    } while (0)

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
// IEEE binary16 conversions in portable integer code (round to nearest even, denormals, overflow to infinity): the host
// compiler need not know _Float16, and the image is the same whatever compiled this file
static inline uint16_t f32_to_f16_rne(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));   // infinity / NaN
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                      // >= 65520 rounds to infinity
    if (x < 0x38800000u) {                                                                        // below 2^-14: denormal half or zero
        if (x < 0x33000000u) return (uint16_t)sign;                                               // below 2^-25: zero
        const int shift = 126 - (int)(x >> 23);                                                   // 14 .. 24: down to units of 2^-24
        const uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        return (uint16_t)(sign | (q + ((rem > half) || (rem == half && (q & 1u)))));
    }
    uint32_t r = (((x >> 23) - 112u) << 10) | ((x & 0x7fffffu) >> 13);
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;          // a carry into the exponent is the correct rounding
    return (uint16_t)(sign | r);
}
static inline float f16_to_f32(uint16_t h)
{
    const uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        float f = (float)m * 5.9604644775390625e-08f;                 // m * 2^-24, exact
        memcpy(&x, &f, 4);
        x |= s;
    } else if (e == 31) {
        x = s | 0x7f800000u | (m << 13);
    } else {
        x = s | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

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
    uint16_t *const dst = reinterpret_cast<uint16_t *>(base);
    for (int mt = 0; mt < mtiles; ++mt)
        for (int c = 0; c < nchunks; ++c)
            for (int tap = 0; tap < kernel; ++tap) {
                uint16_t *grp = dst + (((size_t)mt * nchunks + c) * kernel + tap) * (2 * 64 * 8);
                for (int l = 0; l < 64; ++l) {
                    const int row = mt * 32 + (l & 31);
                    if (row >= c_out) continue;
                    for (int i = 0; i < 8; ++i) {
                        const int ci = c * kConvCK + 8 * (l >> 5) + i;
                        if (ci >= c_in) continue;
                        const float v = ldexpf(w[((long)row * c_in + ci) * kernel + tap], rexp[row]);     // exact
                        const uint16_t hi = f32_to_f16_rne(v);                                           // round to nearest even
                        const uint16_t lo = f32_to_f16_rne((v - f16_to_f32(hi)) * 2048.f);               // residual exact
                        grp[(0 * 64 + l) * 8 + i] = hi;
                        grp[(1 * 64 + l) * 8 + i] = lo;
                    }
                }
            }
    return TTSAMD_OK;
}


// ---- fused ResBlock pair: which (channels, kernel, dilation) have an instantiation, and the image sizes it reads ----------------
extern "C" int ttsamd_resblock_pair_supported(int c, int kernel, int dilation)
{
    return (c == 8 || c == 16 || c == 32 || c == 64 || c == 128) && (kernel == 3 || kernel == 7 || kernel == 11) &&
           (dilation == 1 || dilation == 3 || dilation == 5);
}

// ... with the two-part fp16 images (three-product arithmetic): additionally c = 256
extern "C" int ttsamd_resblock_pair_h2_supported(int c, int kernel, int dilation)
{
    return (ttsamd_resblock_pair_supported(c, kernel, dilation) || c == 256) && (kernel == 3 || kernel == 7 || kernel == 11) &&
           (dilation == 1 || dilation == 3 || dilation == 5);
}

extern "C" size_t ttsamd_resblock_weight_bytes(int c, int kernel)
{
    const int cc = c < 32 ? 32 : c;
    return ttsamd_conv1d_packed_split_bytes(cc, cc, kernel);
}

extern "C" size_t ttsamd_resblock_weight_h2_bytes(int c, int kernel)
{
    const int cc = c < 32 ? 32 : c;
    return ttsamd_conv1d_packed_h2_bytes(cc, cc, kernel);
}

