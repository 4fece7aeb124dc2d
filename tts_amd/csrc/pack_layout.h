// Layout constants of the packed weight images, shared by the kernels (device code) and the host-side pack functions
// (pack_host.cpp: plain C++, no HIP headers — it is also compiled stand-alone under -fsanitize=address,undefined by
// tests/test_host_cpu.py).
#pragma once
#include <cstddef>

#if defined(__HIPCC__)
#define TTSAMD_HD __host__ __device__
#else
#define TTSAMD_HD
#endif

namespace ttsamd {

constexpr int kConvCK = 16;                     // input channels per LDS chunk / per group of the weight images
constexpr int kH2GroupBytes = 2 * 64 * 16;      // one (chunk, tap) group of the two-part fp16 image: 2 parts x 64 lanes x 16 bytes

// header of the row table that follows the fragment groups of a two-part fp16 weight image (conv_kernel_h2.h)
struct H2RowTable {
    int max_row_exp;        // largest row exponent of the image
    int pad[3];
    // then per packed row (mtiles * 32): float scale = 2^e_row, float unscale = 2^-e_row
};

TTSAMD_HD inline size_t conv_h2_table_offset(int c_out, int c_in, int kernel)
{
    const size_t mtiles = (size_t)(c_out + 31) / 32;
    const size_t nchunks = (size_t)(c_in + kConvCK - 1) / kConvCK;
    return (mtiles * nchunks * kernel + 2) * kH2GroupBytes;     // + two zero groups of prefetch slack
}

}  // namespace ttsamd
