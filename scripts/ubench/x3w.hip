// Products-per-output experiment (NOT part of the library build): the dominant conv of the headline step —
// Conv1d C -> C, k = 11 (and 7, 3), dilation 1, split-bf16 arithmetic — with 1-D minimal filtering F(2,3) on tap triples:
// output pairs (y[2p], y[2p+1]) from FOUR products per triple instead of six, i.e. 16 / 12 / 4 MFMA groups per 16-channel chunk
// and 64 output columns instead of 22 / 14 / 6 (k = 11 and 7 are zero-padded to 12 and 9 taps).
//   m0 = (d0 - d2) g0          m1 = (d1 + d2) (g0 + g1 + g2)/2      m2 = (d2 - d1) (g0 - g1 + g2)/2      m3 = (d1 - d3) g2
//   y0 = m0 + m1 + m2          y1 = m1 - m2 - m3
// With A(s) = x[s] - x[s+2], P(s) = x[s] + x[s+1], M(s) = x[s+1] - x[s] the transformed inputs of EVERY triple are shifted
// reads of three staged signals (d_i = x[2p + 3g + i]):  d0-d2 = A(2p+3g), d1+d2 = P(2p+3g+1), d2-d1 = M(2p+3g+1),
// d1-d3 = A(2p+3g+1) — so a tile is transformed once per 16-channel chunk, like the direct kernel stages it once: each signal
// is computed in fp32 from the activated inputs, split exactly into three bf16 parts and stored in LDS in even / odd position
// planes (a B fragment = 32 consecutive pairs = one parity plane, 16-byte stride: conflict free).  Transformed weights are
// computed in fp64 on the host, rounded once to fp32 and split like any weight (ttsamd_conv1d_pack_weights_split of a
// [c_out, c_in, 4 * groups] tensor).  Accuracy of the scheme under sequential fp32 accumulation: scripts/ubench/winograd_emul.py.
//
// Block = 128 output rows x 128 output columns (4 waves, one 32-row m-tile each; 2 pair-tiles of 32 pairs), eight accumulator
// tiles per wave (m0..m3 x 2 pair-tiles), single LDS buffer of 3 signals (41 KB: two blocks per CU), the next chunk's inputs
// requested before the MFMA phase and transformed after it.  Epilogue: output transform, + bias (+ residual), 8-byte stores.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off x3w.hip -o libx3w.so   (driver: x3w_bench.py)
#include "../../tts_amd/csrc/conv_kernel_x3.h"

#include <cstdlib>

namespace ttsamd {
void set_error(const char *, ...) {}
std::atomic<unsigned long long> g_launches{0};
}

namespace x3w {
using namespace ttsamd;

struct Args {
    const float *x;          // [B, C, T]
    const void *wu_split;    // split image of the transformed weights [C, C, 4 * G]
    const float *bias;       // [C] or NULL
    const float *res;        // [B, C, T] or NULL
    float *y;                // [B, C, T]
    int c, t, batch;
    float slope;             // leaky-ReLU slope of the input activation (1 = none)
};

template <int K>
struct Geom {
    static constexpr int kG = (K + 2) / 3;                  // tap triples (K zero-padded to 3 G)
    static constexpr int kSlots = 4 * kG;                   // transformed-weight slots per chunk
    static constexpr int kPad = (K - 1) / 2;
    static constexpr int kBN = 128;                         // output columns per block = 64 pairs
    static constexpr int kXS = kBN + 3 * kG + 1;            // signal positions 0 .. 2*63 + 3(G-1) + 1
    static constexpr int kPlaneMin = (kXS + 1) / 2 + 1;     // per parity plane (+ one dump column)
    static constexpr int kPlaneCols = kPlaneMin + ((11 - kPlaneMin % 16) + 16) % 16;   // = 11 mod 16: the two 8-channel halves of a
                                                            // fragment read land 176 mod 256 bytes apart, as in the direct kernel
    static constexpr int kPlaneBytes = kPlaneCols * 16;
    static constexpr int kSigBytes = 3 * 2 * 2 * kPlaneBytes;   // [part][parity][half]
    static constexpr int kLdsBytes = 3 * kSigBytes;             // A, P, M
    static constexpr int kThreads = 256;
    static constexpr int kItems = 2 * kXS;                  // (half, position)
    static constexpr int kRounds = (kItems + kThreads - 1) / kThreads;
};

template <int K>
__global__ __launch_bounds__(256, 2) void conv_x3w_kernel(const Args a)
{
    using G = Geom<K>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int j = lane & 31;
    const int b = blockIdx.z;
    const int mb = blockIdx.y;                               // 128-row block
    const int t0 = blockIdx.x * G::kBN;
    const int C = a.c, T = a.t;
    const int nchunks = C / 16;
    constexpr int kOob = kConvOob;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)b * C * T, (long)C * T * 4);
    const int row_bytes = T * 4;

    // staging items: (half, position s); x[s], x[s+1], x[s+2] of 8 channels each
    int soff[G::kRounds][3];
    int lplane[G::kRounds];                                  // byte offset inside a (signal, part) block: half, parity, column
#pragma unroll
    for (int r = 0; r < G::kRounds; ++r) {
        const int e = tid + r * G::kThreads;
        const bool live = e < G::kItems;
        const int half = live ? e / G::kXS : 1;
        const int s = live ? e - half * G::kXS : G::kXS - 1;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int gt = t0 - G::kPad + s + d;
            soff[r][d] = (live && gt >= 0 && gt < T) ? (int)(((long)(half * 8) * T + gt) * 4) : kOob;
        }
        // idle lanes of the last round write the dump column of their plane
        lplane[r] = ((s & 1) * 2 + half) * G::kPlaneBytes + (live ? (s >> 1) : G::kPlaneCols - 1) * 16;
    }
    float st[G::kRounds][3][8];
    auto stage_load = [&](int chunk) {
        const int cb = chunk * 16 * row_bytes;
#pragma unroll
        for (int r = 0; r < G::kRounds; ++r)
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int c = 0; c < 8; ++c) st[r][d][c] = ld_buf(rx, soff[r][d] == kOob ? kOob : soff[r][d] + cb + c * row_bytes, 0);
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int r = 0; r < G::kRounds; ++r) {
            float v[3][8];
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int c = 0; c < 8; ++c) v[d][c] = conv_lrelu(st[r][d][c], a.slope);
#pragma unroll
            for (int sig = 0; sig < 3; ++sig) {
                unsigned pw[3][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float s0, s1;
                    if (sig == 0) { s0 = v[0][2 * c] - v[2][2 * c]; s1 = v[0][2 * c + 1] - v[2][2 * c + 1]; }        // A = x[s] - x[s+2]
                    else if (sig == 1) { s0 = v[0][2 * c] + v[1][2 * c]; s1 = v[0][2 * c + 1] + v[1][2 * c + 1]; }   // P = x[s] + x[s+1]
                    else { s0 = v[1][2 * c] - v[0][2 * c]; s1 = v[1][2 * c + 1] - v[0][2 * c + 1]; }                 // M = x[s+1] - x[s]
                    conv_split3x2(s0, s1, pw[0][c], pw[1][c], pw[2][c]);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    u32x4 w;
                    w.x = pw[q][0];
                    w.y = pw[q][1];
                    w.z = pw[q][2];
                    w.w = pw[q][3];
                    *reinterpret_cast<u32x4 *>(lds + sig * G::kSigBytes + q * (4 * G::kPlaneBytes) + lplane[r]) = w;
                }
            }
        }
    };

    const long mtile = (long)mb * 4 + wave;
    const u32x4 *const wp = reinterpret_cast<const u32x4 *>(a.wu_split) + mtile * ((long)nchunks * G::kSlots * 3 * 64) + lane;
    u32x4 a_cur[3], a_nxt[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) a_cur[q] = wp[q * 64];

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][pt][r] = 0.f;

    stage_load(0);
    stage_store();
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        stage_load(c + 1);                                   // past the last chunk: zeros (range check), no traffic
#pragma unroll
        for (int sl = 0; sl < G::kSlots; ++sl) {
            const int g = sl >> 2, i = sl & 3;
            const long nx = (sl + 1 < G::kSlots) ? ((long)c * G::kSlots + sl + 1) : ((long)(c + 1) * G::kSlots);
#pragma unroll
            for (int q = 0; q < 3; ++q) a_nxt[q] = wp[nx * (3 * 64) + q * 64];        // (the image ends with slack groups)
            __builtin_amdgcn_sched_barrier(0);
            constexpr int sigs[4] = {0, 1, 2, 0};
            constexpr int offs[4] = {0, 1, 1, 1};
            const int pos = 3 * g + offs[i];
            const unsigned char *base = lds + sigs[i] * G::kSigBytes + ((pos & 1) * 2 + h) * G::kPlaneBytes + (j + (pos >> 1)) * 16;
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                u32x4 bq[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) bq[q] = *reinterpret_cast<const u32x4 *>(base + q * (4 * G::kPlaneBytes) + pt * 32 * 16);
                constexpr int pa[6] = {2, 1, 0, 1, 0, 0};
                constexpr int pb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t)
                    acc[i][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_cur[pa[t]]),
                                                                         __builtin_bit_cast(bf16x8, bq[pb[t]]), acc[i][pt], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) a_cur[q] = a_nxt[q];
        }
        __syncthreads();                                     // every wave is done with this chunk's signals
        stage_store();
        __syncthreads();
    }

    // output transform + epilogue: y[2p] = m0 + m1 + m2, y[2p+1] = m1 - m2 - m3, + bias (+ residual); one 8-byte store per lane
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    const int row0 = (int)mtile * 32;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int t = t0 + pt * 64 + 2 * j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float bv = a.bias ? a.bias[row] : 0.f;
            f32x2 o;
            o[0] = ((acc[0][pt][r] + acc[1][pt][r]) + acc[2][pt][r]) + bv;
            o[1] = ((acc[1][pt][r] - acc[2][pt][r]) - acc[3][pt][r]) + bv;
            const long off = ((long)b * C + row) * T + t;
            if (t + 1 < T) {
                if (a.res) {
                    const f32x2 rv = *reinterpret_cast<const f32x2 *>(a.res + off);
                    o[0] += rv[0];
                    o[1] += rv[1];
                }
                *reinterpret_cast<f32x2 *>(a.y + off) = o;
            } else if (t < T) {
                a.y[off] = o[0] + (a.res ? a.res[off] : 0.f);
            }
        }
    }
}

// ---- version 2: 8 waves (4 m-tiles x 2 column halves: 128 rows x 256 columns), double-buffered signals (one barrier per chunk),
// the next chunk's transform spread over the slots of the current one (loads at slot 0; A at 1/3, P at 1/2, M at 2/3 of the slots),
// the few items beyond one full round handled by wave 0 alone behind a wave-uniform branch.
template <int K>
struct Geom2 {
    static constexpr int kG = (K + 2) / 3;
    static constexpr int kSlots = 4 * kG;
    static constexpr int kPad = (K - 1) / 2;
    static constexpr int kBN = 256;                         // output columns per block = 128 pairs
    static constexpr int kXS = kBN + 3 * kG + 1;
    static constexpr int kPlaneCols = (kXS + 1) / 2 + 1;
    static constexpr int kPlaneBytes = kPlaneCols * 16;
    static constexpr int kSigBytes = 3 * 2 * 2 * kPlaneBytes;
    static constexpr int kBufBytes = 3 * kSigBytes;
    static constexpr int kLdsBytes = 2 * kBufBytes;
    static constexpr int kThreads = 512;
    static constexpr int kItems = 2 * kXS;
    static constexpr int kExtra = kItems - kThreads;        // items of the second, partial round (wave 0 only)
    static_assert(kExtra > 0 && kExtra <= 64, "one full round + a partial one on wave 0");
};

template <int K>
__global__ __launch_bounds__(512, 2) void conv_x3w2_kernel(const Args a)
{
    using G = Geom2<K>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5;
    const int j = lane & 31;
    const int b = blockIdx.z;
    const int mb = blockIdx.y;
    const int t0 = blockIdx.x * G::kBN;
    const int C = a.c, T = a.t;
    const int nchunks = C / 16;
    constexpr int kOob = kConvOob;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)b * C * T, (long)C * T * 4);
    const int row_bytes = T * 4;

    // item of round 0: e = tid; item of round 1 (wave 0, lanes < kExtra): e = kThreads + lane
    int soff[2][3], lplane[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int e = r == 0 ? tid : G::kThreads + lane;
        const bool live = e < G::kItems;
        const int half = live ? e / G::kXS : 1;
        const int s = live ? e - half * G::kXS : G::kXS - 1;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int gt = t0 - G::kPad + s + d;
            soff[r][d] = (live && gt >= 0 && gt < T) ? (int)(((long)(half * 8) * T + gt) * 4) : kOob;
        }
        lplane[r] = (half * 2 + (s & 1)) * G::kPlaneBytes + (live ? (s >> 1) : G::kPlaneCols - 1) * 16;
    }
    float st[3][8], st1[3][8];
    auto load_item = [&](int r, float(&dst)[3][8], int chunk) {
        const int cb = chunk * 16 * row_bytes;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int c = 0; c < 8; ++c) dst[d][c] = ld_buf(rx, soff[r][d] == kOob ? kOob : soff[r][d] + cb + c * row_bytes, 0);
    };
    auto act_item = [&](float(&v)[3][8]) {
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int c = 0; c < 8; ++c) v[d][c] = conv_lrelu(v[d][c], a.slope);
    };
    auto store_sig = [&](int sig, const float(&v)[3][8], unsigned char *buf, int lp) {
        unsigned pw[3][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float s0, s1;
            if (sig == 0) { s0 = v[0][2 * c] - v[2][2 * c]; s1 = v[0][2 * c + 1] - v[2][2 * c + 1]; }
            else if (sig == 1) { s0 = v[0][2 * c] + v[1][2 * c]; s1 = v[0][2 * c + 1] + v[1][2 * c + 1]; }
            else { s0 = v[1][2 * c] - v[0][2 * c]; s1 = v[1][2 * c + 1] - v[0][2 * c + 1]; }
            conv_split3x2(s0, s1, pw[0][c], pw[1][c], pw[2][c]);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            u32x4 w;
            w.x = pw[q][0];
            w.y = pw[q][1];
            w.z = pw[q][2];
            w.w = pw[q][3];
            *reinterpret_cast<u32x4 *>(buf + sig * G::kSigBytes + q * (4 * G::kPlaneBytes) + lp) = w;
        }
    };

    const long mtile = (long)mb * 4 + wm;
    const u32x4 *const wp = reinterpret_cast<const u32x4 *>(a.wu_split) + mtile * ((long)nchunks * G::kSlots * 3 * 64) + lane;
    u32x4 a_cur[3], a_nxt[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) a_cur[q] = wp[q * 64];
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][pt][r] = 0.f;

    // chunk 0 -> buffer 0
    load_item(0, st, 0);
    act_item(st);
#pragma unroll
    for (int sig = 0; sig < 3; ++sig) store_sig(sig, st, lds, lplane[0]);
    if (wave == 0) {
        load_item(1, st1, 0);
        act_item(st1);
#pragma unroll
        for (int sig = 0; sig < 3; ++sig) store_sig(sig, st1, lds, lplane[1]);
    }
    __syncthreads();
    constexpr int sA = G::kSlots / 3, sP = G::kSlots / 2, sM = (2 * G::kSlots) / 3;
    for (int c = 0; c < nchunks; ++c) {
        const unsigned char *cur = lds + (c & 1) * G::kBufBytes;
        unsigned char *const nxt = lds + ((c + 1) & 1) * G::kBufBytes;
#pragma unroll
        for (int sl = 0; sl < G::kSlots; ++sl) {
            const int g = sl >> 2, i = sl & 3;
            const long nx = (sl + 1 < G::kSlots) ? ((long)c * G::kSlots + sl + 1) : ((long)(c + 1) * G::kSlots);
#pragma unroll
            for (int q = 0; q < 3; ++q) a_nxt[q] = wp[nx * (3 * 64) + q * 64];
            if (sl == 0) {
                load_item(0, st, c + 1);
                if (wave == 0) load_item(1, st1, c + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (sl == sA) { act_item(st); store_sig(0, st, nxt, lplane[0]); }
            if (sl == sP) store_sig(1, st, nxt, lplane[0]);
            if (sl == sM) store_sig(2, st, nxt, lplane[0]);
            constexpr int sigs[4] = {0, 1, 2, 0};
            constexpr int offs[4] = {0, 1, 1, 1};
            const int pos = 3 * g + offs[i];
            const unsigned char *base = cur + sigs[i] * G::kSigBytes + (h * 2 + (pos & 1)) * G::kPlaneBytes + (wn * 64 + j + (pos >> 1)) * 16;
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                u32x4 bq[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) bq[q] = *reinterpret_cast<const u32x4 *>(base + q * (4 * G::kPlaneBytes) + pt * 32 * 16);
                constexpr int pa[6] = {2, 1, 0, 1, 0, 0};
                constexpr int pb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t)
                    acc[i][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_cur[pa[t]]),
                                                                         __builtin_bit_cast(bf16x8, bq[pb[t]]), acc[i][pt], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) a_cur[q] = a_nxt[q];
        }
        if (wave == 0) {      // the partial second round of the next chunk (a few dozen items)
            act_item(st1);
#pragma unroll
            for (int sig = 0; sig < 3; ++sig) store_sig(sig, st1, nxt, lplane[1]);
        }
        __syncthreads();
    }

    using f32x2 = __attribute__((ext_vector_type(2))) float;
    const int row0 = (int)mtile * 32;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int t = t0 + wn * 128 + pt * 64 + 2 * j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float bv = a.bias ? a.bias[row] : 0.f;
            f32x2 o;
            o[0] = ((acc[0][pt][r] + acc[1][pt][r]) + acc[2][pt][r]) + bv;
            o[1] = ((acc[1][pt][r] - acc[2][pt][r]) - acc[3][pt][r]) + bv;
            const long off = ((long)b * C + row) * T + t;
            if (t + 1 < T) {
                if (a.res) {
                    const f32x2 rv = *reinterpret_cast<const f32x2 *>(a.res + off);
                    o[0] += rv[0];
                    o[1] += rv[1];
                }
                *reinterpret_cast<f32x2 *>(a.y + off) = o;
            } else if (t < T) {
                a.y[off] = o[0] + (a.res ? a.res[off] : 0.f);
            }
        }
    }
}

template <int K>
static int launch2(const Args &a, hipStream_t st)
{
    using G = Geom2<K>;
    auto kern = conv_x3w2_kernel<K>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::kLdsBytes) != hipSuccess)
        return -2;
    hipLaunchKernelGGL(kern, dim3((a.t + G::kBN - 1) / G::kBN, a.c / 128, a.batch), dim3(G::kThreads), G::kLdsBytes, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int K>
static int launch(const Args &a, hipStream_t st)
{
    using G = Geom<K>;
    auto kern = conv_x3w_kernel<K>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::kLdsBytes) != hipSuccess)
        return -2;
    hipLaunchKernelGGL(kern, dim3((a.t + G::kBN - 1) / G::kBN, a.c / 128, a.batch), dim3(G::kThreads), G::kLdsBytes, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace x3w

// C must be a multiple of 128, T even.  wu_split: ttsamd_conv1d_pack_weights_split image of the transformed weights
// [C, C, 4 * ceil(k / 3)] (slot 4 g + i = transform i of tap triple g).
extern "C" int x3w_conv(const float *x, const void *wu_split, const float *bias, const float *res, float *y, int c, int t, int batch,
                        int kernel, float slope, void *stream)
{
    if (c % 128 || t % 2) return -1;
    x3w::Args a{x, wu_split, bias, res, y, c, t, batch, slope};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (getenv("X3W_V2")) {
        switch (kernel) {
            case 11: return x3w::launch2<11>(a, st);
            case 7: return x3w::launch2<7>(a, st);
        }
        return -1;
    }
    switch (kernel) {
        case 11: return x3w::launch<11>(a, st);
        case 7: return x3w::launch<7>(a, st);
        case 3: return x3w::launch<3>(a, st);
    }
    return -1;
}
