// Sanitizer pass of the host-side pack / policy code and of the MAS oracle: built by tests/test_host_cpu.py with
//   cc  -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -c ../../oracle/mas_oracle.c
//   c++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all sanitize_driver.cpp ../../tts_amd/csrc/pack_host.cpp mas_oracle.o
// and run on the CPU.  It packs weights of awkward shapes (channel counts that are not multiples of the tile sizes, kernel sizes
// 1..31, rows of zeros / denormals / huge magnitudes) into exactly-sized heap buffers — any write past ttsamd_*_bytes() trips the
// address sanitizer —, checks every byte of the images is either written or zero padding, unpacks the two-part fp16 image back to
// the weights, walks the policy queries over their whole argument range, and runs the C restatement of maximum_path_c on ragged
// random problems (ties, t_x == 1, t_x == t_y) with the path checked for monotonicity.
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/tts_amd.h"
#include "../../tts_amd/csrc/pack_layout.h"

namespace ttsamd {
static char g_err[512];
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ttsamd

extern "C" void mas_oracle_c(int32_t *paths, float *values, const int32_t *t_xs, const int32_t *t_ys, int b, int t_x_max, int t_y_max,
                             float max_neg_val);      // oracle/mas_oracle.c

static unsigned long long rng_state = 88172645463325252ull;
static double urand()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}
#define REQUIRE(c)                                                            \
    do {                                                                      \
        if (!(c)) {                                                           \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);    \
            return 1;                                                         \
        }                                                                     \
    } while (0)

static float half_to_float(uint16_t h)
{
    const uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
    if (e == 0) return (s ? -1.f : 1.f) * (float)m * 5.9604644775390625e-08f;
    if (e == 31) return m ? NAN : (s ? -INFINITY : INFINITY);
    return (s ? -1.f : 1.f) * ldexpf((float)(m | 0x400u), (int)e - 25);
}

int main()
{
    const int shapes[][3] = {{1, 1, 1}, {32, 16, 3}, {33, 17, 7}, {80, 192, 1}, {100, 40, 11}, {64, 64, 31}, {5, 300, 2}, {129, 31, 5}};
    for (const auto &sh : shapes) {
        const int co = sh[0], ci = sh[1], k = sh[2];
        std::vector<float> w((size_t)co * ci * k);
        for (size_t i = 0; i < w.size(); ++i) w[i] = (float)((urand() - 0.5) * std::pow(10.0, urand() * 8 - 6));
        if (co > 2) {
            for (int i = 0; i < ci * k; ++i) w[(size_t)1 * ci * k + i] = 0.f;                       // an all-zero row
            for (int i = 0; i < ci * k; ++i) w[(size_t)2 * ci * k + i] = (i & 1) ? 1e-41f : -3e-39f;  // a denormal row
            w[0] = 3e38f;                                                                            // a huge weight
        }
        // fp32 fragment image (kernel sizes whose k-steps fill groups of four only)
        if (((16 / 2) * k) % 4 == 0) {
            const size_t n = ttsamd_conv1d_packed_floats(co, ci, k);
            float *img = (float *)malloc(n * sizeof(float));
            REQUIRE(ttsamd_conv1d_pack_weights(img, w.data(), co, ci, k) == TTSAMD_OK);
            double s0 = 0, s1 = 0;
            for (size_t i = 0; i < n; ++i) s0 += (double)img[i];
            for (float v : w) s1 += (double)v;
            REQUIRE(std::fabs(s0 - s1) <= 1e-6 * std::fabs(s1) + 1e-30);      // every weight exactly once, zeros elsewhere
            free(img);
        }
        {   // split-bf16 image: parts of a weight sum back to it exactly (above bf16's denormal range)
            const size_t nb = ttsamd_conv1d_packed_split_bytes(co, ci, k);
            uint16_t *img = (uint16_t *)malloc(nb);
            REQUIRE(ttsamd_conv1d_pack_weights_split(img, w.data(), co, ci, k) == TTSAMD_OK);
            const int nch = (ci + 15) / 16;
            for (int row = 0; row < co; row += 7)
                for (int c = 0; c < ci; c += 5)
                    for (int tap = 0; tap < k; ++tap) {
                        const size_t grp = (((size_t)(row / 32) * nch + c / 16) * k + tap) * (3 * 64 * 8);
                        const int l = (row % 32) + 32 * ((c % 16) / 8), i = c % 8;
                        float sum = 0.f;
                        for (int q = 2; q >= 0; --q) {
                            const uint32_t u = (uint32_t)img[grp + (q * 64 + l) * 8 + i] << 16;
                            float f;
                            memcpy(&f, &u, 4);
                            sum += f;
                        }
                        const float v = w[((size_t)row * ci + c) * k + tap];
                        REQUIRE(sum == v || std::fabs(v) < 1e-30f);
                    }
            free(img);
        }
        {   // two-part fp16 image: hi + lo / 2048 reproduces w * 2^e_row to 2^-22 of the row maximum's binade; table consistent
            const size_t nb = ttsamd_conv1d_packed_h2_bytes(co, ci, k);
            unsigned char *img = (unsigned char *)malloc(nb);
            REQUIRE(ttsamd_conv1d_pack_weights_h2(img, w.data(), co, ci, k) == TTSAMD_OK);
            const size_t off = ttsamd::conv_h2_table_offset(co, ci, k);
            const ttsamd::H2RowTable *hdr = (const ttsamd::H2RowTable *)(img + off);
            const float *tab = (const float *)(hdr + 1);
            const uint16_t *parts = (const uint16_t *)img;
            const int nch = (ci + 15) / 16;
            int emax = -1000;
            for (int row = 0; row < co; ++row) {
                const float sc = tab[2 * row], us = tab[2 * row + 1];
                REQUIRE(sc > 0.f && sc * us == 1.f);
                int e;
                frexpf(sc, &e);
                if (e - 1 > emax) emax = e - 1;
                float mx = 0.f;
                for (int i = 0; i < ci * k; ++i) mx = std::fmax(mx, std::fabs(w[(size_t)row * ci * k + i]));
                if (mx > 0.f && sc < 8e37f && sc > 2e-38f) REQUIRE(mx * sc >= 8192.f * 0.999f && mx * sc < 16384.f * 1.001f);
                for (int c = 0; c < ci; c += 3)
                    for (int tap = 0; tap < k; ++tap) {
                        const size_t grp = (((size_t)(row / 32) * nch + c / 16) * k + tap) * (2 * 64 * 8);
                        const int l = (row % 32) + 32 * ((c % 16) / 8), i = c % 8;
                        const float hi = half_to_float(parts[grp + (0 * 64 + l) * 8 + i]), lo = half_to_float(parts[grp + (1 * 64 + l) * 8 + i]);
                        const double want = (double)w[((size_t)row * ci + c) * k + tap] * (double)sc;
                        REQUIRE(std::isfinite(hi) && std::isfinite(lo));
                        REQUIRE(std::fabs((double)hi + (double)lo / 2048.0 - want) <= std::fabs(want) * 2.4e-7 + 3e-11);
                    }
            }
            REQUIRE(hdr->max_row_exp == emax);
            free(img);
        }
    }
    // argument checks return errors, never touch memory
    REQUIRE(ttsamd_conv1d_pack_weights_h2(nullptr, nullptr, 4, 4, 3) == TTSAMD_ERR_INVALID && ttsamd::g_err[0]);
    REQUIRE(ttsamd_conv1d_pack_weights_split(nullptr, nullptr, 0, 4, 3) == TTSAMD_ERR_INVALID);
    REQUIRE(ttsamd_conv1d_packed_h2_bytes(0, 1, 1) == 0 && ttsamd_conv1d_packed_split_bytes(1, -1, 1) == 0 && ttsamd_conv1d_packed_floats(1, 1, 0) == 0);
    int tuned = 0, supported = 0, fused = 0;
    for (int k = -2; k < 40; ++k)
        for (int d = -2; d < 40; ++d) {
            tuned += ttsamd_conv1d_tuned(k, d);
            supported += ttsamd_conv1d_supported(k, d);
            for (int c = -1; c < 300; c += 1) fused += ttsamd_resblock_pair_supported(c, k, d);
        }
    REQUIRE(tuned == 3 + 3 * 3 + 1 && supported == 31 * 27 && fused == 5 * 3 * 3);
    REQUIRE(ttsamd_resblock_weight_bytes(8, 3) == ttsamd_conv1d_packed_split_bytes(32, 32, 3));
    REQUIRE(ttsamd_resblock_weight_h2_bytes(128, 11) == ttsamd_conv1d_packed_h2_bytes(128, 128, 11));

    // the C restatement of maximum_path_c (oracle/mas_oracle.c) on ragged problems: exactly-sized buffers, path monotone and complete
    for (int rep = 0; rep < 6; ++rep) {
        const int B = 3, TX = 5 + rep * 9, TY = TX + rep * 13;
        std::vector<float> val((size_t)B * TX * TY);
        std::vector<int> path((size_t)B * TX * TY, 0), txs(B), tys(B);
        for (auto &v : val) v = (rep & 1) ? (float)(int)(urand() * 3) : (float)(urand() - 0.5);      // odd reps: tie-heavy integer grids
        for (int b = 0; b < B; ++b) {
            txs[b] = b == 0 ? TX : (b == 1 ? 1 : 1 + (int)(urand() * (TX - 1)));
            tys[b] = b == 2 ? txs[b] : txs[b] + (int)(urand() * (TY - txs[b] + 1));
            if (tys[b] > TY) tys[b] = TY;
        }
        mas_oracle_c(path.data(), val.data(), txs.data(), tys.data(), B, TX, TY, -1e9f);
        for (int b = 0; b < B; ++b) {
            int prev = -1;
            for (int y = 0; y < tys[b]; ++y) {
                int cnt = 0, at = -1;
                for (int x = 0; x < TX; ++x)
                    if (path[((size_t)b * TX + x) * TY + y]) { ++cnt; at = x; }
                REQUIRE(cnt == 1 && at < txs[b] && (at == prev || at == prev + 1 || prev < 0));
                if (y == 0) REQUIRE(at == 0);
                prev = at;
            }
            REQUIRE(prev == txs[b] - 1);
        }
    }
    printf("sanitize_driver: ok\n");
    return 0;
}
