// Conv1d as implicit GEMM on the bf16 MFMA of gfx950 (v_mfma_f32_32x32x16_bf16) with BOTH fp32 operands split three
// ways into bf16 (x = x1 + x2 + x3, each part the round-to-nearest bf16 of what is left) and the six products whose
// magnitude is >= 2^-16 |w x| accumulated in fp32:
//
//     w x  ~=  w1 x1 + (w1 x2 + w2 x1) + (w1 x3 + w2 x2 + w3 x1)
//
// The split itself is exact (24 significand bits = 3 x 8: w1+w2+w3 == w, x1+x2+x3 == x bit for bit, above bf16's
// denormal range).  The three dropped products (w2 x3, w3 x2, w3 x3) sum to at most 2^-24.2 |w x| in the worst case and
// 2^-27.4 |w x| RMS (measured over 2 M random pairs, CPU emulation) — about ONE fp32 rounding of the product in the worst
// case, far below it on average; accumulation is fp32 as in the fp32-input MFMA path (conv_kernel.h).  Same arithmetic
// class, same parity tolerances (tests/test_conv_gpu.py runs adversarial operands — maximal bf16 residuals, K = 2816
// cancellation, activations below bf16's normal range — against an fp64 conv), but the bf16 matrix pipe runs 16x the
// fp32-input rate, so six of its instructions cost 6/16 of the one they replace.
//
//   * A (weights): split and packed at load time as [m-tile][chunk][tap][part][64 lanes][8 bf16]: one 16-byte load per
//     lane per (tap, part) straight from L2, prefetched one tap ahead.  Lane l holds row l%32, channels 8*(l/32)..+7.
//   * B (activations): staged per 16-channel chunk; a thread owns (column, 8-channel half): 8 coalesced fp32 loads,
//     activation / mask, split, three ds_write_b128 into [part][half][column][8 ch] bf16.  A fragment is then ONE
//     ds_read_b128 (lane -> column, half-wave -> 8-channel half = its own plane): consecutive columns are consecutive
//     16-byte slots, so the 16 lanes the LDS serves per cycle ({0-3,12-15,20-27}, ...) hit 16 different slots —
//     conflict free — and tap / dilation are immediates on the column index.  (TTSAMD_X3_PLANAR=0 builds the earlier
//     [part][column][16 ch] image, whose 32-byte rows put two lanes of each group on the same slot: 2-way conflicts.)
//   * one MFMA K-step = 16 channels of one tap; any kernel size works (no pairing constraint).
//   * accumulators, residual folding and the fused epilogues are the fp32 path's (conv_acc_init / conv_epilogue).
#pragma once
#include <cstdlib>
#include "conv_kernel.h"

// Wave-tile arrangement <MI,NI,WM,WN> of the unpaired-row modes (build-time so that variants can be A/B-ed through
// TTSAMD_LIB_PATH): 128-row blocks and 64-row blocks.  See conv1d_x3_launch_tiles for the measurements.
#ifndef TTSAMD_X3_CFG128
#define TTSAMD_X3_CFG128 1, 4, 4, 1
#endif
#ifndef TTSAMD_X3_CFG64
#define TTSAMD_X3_CFG64 1, 4, 2, 2
#endif
namespace ttsamd {

// The three-product kernel adds row / chunk offsets to a lane's byte offset without a select (an invalid lane starts from 2^31, chunks
// up to three beyond c_in are requested ahead and must read zeros through the range check): the sums must stay below 2^32, i.e. the
// per-item slab plus 64 rows below 2 GiB.  Larger slabs (never seen: a per-item [C, T] tensor of ~2 GiB) take the six-product kernel.
inline bool conv_h2_offsets_ok(const ttsamd_conv1d_args &a) { return ((long)(a.c_in + 64) * a.x_rstride + a.t_in) * 4 < 0x7FFFFFF0L; }
extern long g_conv_small_grid_blocks;       // conv.hip (default 128): up to this many 128x128-class blocks a launch takes the small-grid tiles
#define kConvSmallGridBlocks g_conv_small_grid_blocks
constexpr long kConvWaveTileBlocks = 1024;  // up to this many 32x32 tiles those kernels run a wave per tile and K slice
extern int g_conv_small_grid;               // conv.hip: 0 = off, 1 = small tiles, 2 = + K-split groups, 3 = + conv_kernel_x3s.h kernels, 4 = + conv_kernel_x3o.h (default)
extern long g_conv_h2_mid_min;              // conv.hip (default 48): from this many 128x128-class blocks up to the small-grid limit a NORMAL conv
                                            // with the two-part fp16 image takes the three-product kernel on 128 x 64 tiles (TTSAMD_H2_MID_MIN)
}
#ifndef TTSAMD_X3_PLANAR
#define TTSAMD_X3_PLANAR 1
#endif
// The conv_kernel_x3s.h kernels for EVERY kernel size / dilation on small grids (the waveform decoder's 512/256-channel stages of
// a single utterance, conv_pre, HiFiGAN-v2's k = 7 / 11 ResBlocks), not only k <= 5 at dilation 1.  Measured round 3, same box:
// Glow-TTS + HiFiGAN-v2 sentence 3.49 -> 3.28 ms, VITS B = 1 request +-0 (4.74 / 4.79 ms).  TTSAMD_X3S_ALL=0 builds the
// round-2 selection.
#ifndef TTSAMD_X3S_ALL
#define TTSAMD_X3S_ALL 1
#endif

// Staging pipeline of the large-grid kernel (build-time, A/B through TTSAMD_LIB_PATH):
//   0  round-1/2 form: the next chunk's loads are issued behind `if (it + 1 < niter)` at the top of the iteration and
//      converted / written to LDS after the tap loop.  Behind that wave-uniform branch hipcc cannot count the requests in
//      flight and makes the FIRST MFMA of every iteration wait for all of them (`s_waitcnt vmcnt(3)`: an exposed HBM round
//      trip per 16-channel chunk — the device assembly shows it).
//   1  the same schedule with the loads issued unconditionally (a chunk index past c_in reads zeros through the buffer
//      range check without touching memory): exact wait counts.
//   2  (default) the staging work is spread over the taps: item i of the next chunk is requested at the start of tap
//      LT(i) — after that tap's weight requests, so the in-order vmcnt does not force it home before tap LT(i) + 2 — and
//      converted, split and written into the OTHER LDS buffer at the start of tap LT(i) + 2, between that tap's MFMAs.
//      No phase of the iteration is VALU-only or memory-only any more.
#ifndef TTSAMD_X3_PIPE
#define TTSAMD_X3_PIPE 2
#endif

namespace ttsamd {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int K, int D, int MI, int NI, int WM, int WN>
struct ConvGeomX3 {
    static constexpr int kThreads = 64 * WM * WN;
    static constexpr int kBM = 32 * MI * WM;
    static constexpr int kBN = 32 * NI * WN;
    static constexpr int kHalo = (K - 1) * D;
    static constexpr int kXW = kBN + kHalo;                  // staged columns
    static constexpr int kXWp = kXW + 1;                     // + one dump column per plane: where the idle lanes of the last
                                                             // staging round write (no branch around the split / LDS writes)
    static constexpr int kPartBytes = kXWp * 32;             // [half][column][8 ch] bf16 (planar) / [column][16 ch]
    static constexpr int kBufBytes = 3 * kPartBytes;
    static constexpr int kItems = 2 * kXW;                   // (column, 8-channel half) work items per chunk
    static constexpr int kNStage = (kItems + kThreads - 1) / kThreads;
    static constexpr size_t kLdsBytes = (size_t)2 * kBufBytes;
    static constexpr int kOcc = 2;
};

__device__ __forceinline__ void conv_split3(float x, unsigned &p1, unsigned &p2, unsigned &p3)
{
    const __bf16 a1 = (__bf16)x;                 // v_cvt_pk_bf16_f32: round to nearest even
    const float r1 = x - (float)a1;              // exact
    const __bf16 a2 = (__bf16)r1;
    const float r2 = r1 - (float)a2;             // exact
    const __bf16 a3 = (__bf16)r2;
    p1 = __builtin_bit_cast(unsigned short, a1);
    p2 = __builtin_bit_cast(unsigned short, a2);
    p3 = __builtin_bit_cast(unsigned short, a3);
}

// Split TWO values per conversion instruction and keep the parts packed — the same round-to-nearest-even conversions and
// exact residuals as conv_split3, bit for bit, without the 12 v_or_b32_sdwa per 8 values that re-pack separately converted
// halves.  Default since round 3 (TTSAMD_SPLIT_PAIRS=0 builds the one-value-at-a-time form for A/B).
#ifndef TTSAMD_SPLIT_PAIRS
#define TTSAMD_SPLIT_PAIRS 1
#endif
__device__ __forceinline__ void conv_split3x2(float x0, float x1, unsigned &w1, unsigned &w2, unsigned &w3)
{
    using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    const f32x2 v = {x0, x1};
    const unsigned u1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 r1 = {x0 - __builtin_bit_cast(float, u1 << 16), x1 - __builtin_bit_cast(float, u1 & 0xffff0000u)};       // exact
    const unsigned u2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
    const f32x2 r2 = {r1[0] - __builtin_bit_cast(float, u2 << 16), r1[1] - __builtin_bit_cast(float, u2 & 0xffff0000u)};  // exact
    w1 = u1;
    w2 = u2;
    w3 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

// KS > 1 (small-grid launches only): the block carries KS wave groups of WM x WN waves; group g reduces the channel chunks
// g, g + KS, ... into its own accumulators through its own LDS double buffer (the serial K loop of a block — what a launch
// of a few blocks is bound by — becomes KS times shorter), and group 0 adds the partial tiles of groups 1..KS-1 (in that
// order: deterministic) from LDS before the epilogue.
template <int K, int D, int MI, int NI, int WM, int WN, int MODE, int KS = 1>
__global__ __launch_bounds__(64 * WM * WN * KS, (KS > 1 ? (WM * WN * KS) / 4 : ConvGeomX3<K, D, MI, NI, WM, WN>::kOcc)) void conv1d_x3_kernel(const ttsamd_conv1d_args a)
{
    using G = ConvGeomX3<K, D, MI, NI, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char xs3_all[];  // per group: [2][3 parts][2 halves][XW][8 ch] bf16
    const int grp = (KS > 1) ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / G::kThreads) : 0;
    unsigned char *const xs3 = xs3_all + (size_t)grp * G::kLdsBytes;

    const int tid = (int)threadIdx.x - grp * G::kThreads;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR): row offsets become scalar soffsets, no waterfall loops
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int h = lane >> 5;   // 8-channel half of the K-step / row group of D
    const int j = lane & 31;   // column inside a 32-wide N tile
    const ConvTile tile = conv_tile_of_block();
    const int b = tile.b;
    const int mb = tile.mb;
    const int t0 = tile.nb * G::kBN;
    const int nchunks = (a.c_in + kConvCK - 1) / kConvCK;
#ifdef TTSAMD_PHASE_CLOCKS
    // debug build only (scripts/phase_clocks.py): shader-clock stamps of one mid-grid block go to a.y2 (unused by NORMAL)
    long long pc[5];
    pc[0] = clock64();
    const long long prt0 = wall_clock64();
#endif

    constexpr int kOob = kConvOob;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)b * a.x_bstride, ((long)(a.c_in - 1) * a.x_rstride + a.t_in) * 4);

    // staged item i of this thread: (column, half); offset of its first channel inside a chunk, kOob when the column is
    // outside [0, t_in) or the slot is padding; channels >= c_in fall outside the slab and read as 0
    int soff[G::kNStage];
    float smask[G::kNStage];
#pragma unroll
    for (int i = 0; i < G::kNStage; ++i) {
        const int e = tid + i * G::kThreads;
        const int half = e / G::kXW;
        const int col = e - half * G::kXW;
        const int gt = t0 - a.pad_left + col;
        const bool ok = (e < G::kItems) && (gt >= 0) && (gt < a.t_in);
        soff[i] = ok ? (int)(((long)(half * 8) * a.x_rstride + gt) * 4) : kOob;
        smask[i] = 1.f;
    }
    if (a.in_mask) {
        const __amdgpu_buffer_rsrc_t rm = make_rsrc(a.in_mask + (long)b * a.t_in, (long)a.t_in * 4);
#pragma unroll
        for (int i = 0; i < G::kNStage; ++i) {
            const int e = tid + i * G::kThreads;
            const int col = e - (e / G::kXW) * G::kXW;
            const int gt = t0 - a.pad_left + col;
            smask[i] = ld_buf(rm, (gt >= 0 && gt < a.t_in) ? gt * 4 : kOob, 0);
        }
    }
    const int row_bytes = (int)a.x_rstride * 4;
    float st[G::kNStage][8];
    auto stage_load_item = [&](int i, int chunk) {
        const int cb = chunk * kConvCK * row_bytes;
#pragma unroll
        for (int c = 0; c < 8; ++c) st[i][c] = ld_buf(rx, soff[i] == kOob ? kOob : soff[i] + cb + c * row_bytes, 0);
    };
    auto stage_store_item = [&](int i, unsigned char *buf) {
        // straight-line: idle lanes (e >= kItems, last round) loaded zeros through the range check and write them into
        // the dump column of plane 1
        const int e = tid + i * G::kThreads;
        const int half = (e < G::kItems) ? e / G::kXW : 1;
        const int col = (e < G::kItems) ? e - half * G::kXW : G::kXW;
#if TTSAMD_SPLIT_PAIRS
        unsigned pw[3][4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            conv_split3x2(conv_in_act(st[i][2 * c] * smask[i], a.in_act, a.in_slope),
                          conv_in_act(st[i][2 * c + 1] * smask[i], a.in_act, a.in_slope), pw[0][c], pw[1][c], pw[2][c]);
#else
        unsigned p[3][8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            conv_split3(conv_in_act(st[i][c] * smask[i], a.in_act, a.in_slope), p[0][c], p[1][c], p[2][c]);
#endif
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            u32x4 w;
#if TTSAMD_SPLIT_PAIRS
            w.x = pw[q][0];
            w.y = pw[q][1];
            w.z = pw[q][2];
            w.w = pw[q][3];
#else
            w.x = p[q][0] | (p[q][1] << 16);
            w.y = p[q][2] | (p[q][3] << 16);
            w.z = p[q][4] | (p[q][5] << 16);
            w.w = p[q][6] | (p[q][7] << 16);
#endif
#if TTSAMD_X3_PLANAR
            *reinterpret_cast<u32x4 *>(buf + q * G::kPartBytes + half * (G::kXWp * 16) + col * 16) = w;
#else
            *reinterpret_cast<u32x4 *>(buf + q * G::kPartBytes + col * 32 + half * 16) = w;
#endif
        }
    };
    auto stage_load = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < G::kNStage; ++i) stage_load_item(i, chunk);
    };
    auto stage_store = [&](unsigned char *buf) {
#pragma unroll
        for (int i = 0; i < G::kNStage; ++i) stage_store_item(i, buf);
    };
    // pipelined staging (TTSAMD_X3_PIPE == 2): item i is requested at tap kLT(i) and converted at tap kLT(i) + 2 (K = "after the taps")
    // (K < 7: too few taps to spread over — measured same-box, the post-loop form with unconditional requests is as fast or faster)
    constexpr bool kPipe = (TTSAMD_X3_PIPE == 2) && KS == 1 && K >= 7;
    auto lt_of = [](int i) constexpr { return (i * K) / G::kNStage; };

    f32x16 acc[MI][NI];

    // A stream of this wave: m-tile (blockIdx.y*WM + wm)*MI + mi, [chunk][tap][part][64 lanes] x 16 bytes
    const u32x4 *wp[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const long mtile = ((long)mb * WM + wm) * MI + mi;
        wp[mi] = reinterpret_cast<const u32x4 *>(a.w_split) + mtile * ((long)nchunks * K * 3 * 64) + lane;
    }
    u32x4 a_cur[MI][3], a_nxt[MI][3];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int q = 0; q < 3; ++q) a_cur[mi][q] = wp[mi][(KS > 1 && grp < nchunks ? (long)grp * K * (3 * 64) : 0) + q * 64];

    const int niter = (nchunks + KS - 1) / KS;            // chunk steps of every group (a group's last step may be empty)
    stage_load(grp);                                       // chunks beyond c_in read as zeros (buffer range check)
    bool folded = false;
    if (KS == 1 || grp == 0) {
        folded = conv_acc_init<MODE, MI, NI, WM, WN>(acc, a, b, mb, t0, wm, wn, h, j);
    } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    }
    stage_store(xs3);
    __syncthreads();
#ifdef TTSAMD_PHASE_CLOCKS
    pc[1] = clock64();
#endif

#if TTSAMD_X3_PLANAR
    constexpr int kColBytes = 16;
    const int bbyte = h * (G::kXWp * 16) + (wn * (32 * NI) + j) * 16;  // this lane's fragment inside a part, tap 0, ni 0
#else
    constexpr int kColBytes = 32;
    const int bbyte = (wn * (32 * NI) + j) * 32 + h * 16;
#endif
    for (int it = 0; it < niter; ++it) {
        const int c = grp + it * KS;
        const unsigned char *cur = xs3 + (it & 1) * G::kBufBytes + bbyte;
        unsigned char *const nxt = xs3 + ((it + 1) & 1) * G::kBufBytes;
        if constexpr (!kPipe) {
#if TTSAMD_X3_PIPE == 0
            if (it + 1 < niter) stage_load(c + KS);
#else
            stage_load(c + KS);      // past the last chunk: zeros from the buffer range check, no memory traffic
#endif
        }
        if (KS == 1 || c < nchunks) {
#pragma unroll
            for (int tap = 0; tap < K; ++tap) {
                // next weights: the next tap, or the first tap of this group's next chunk (the packed image ends with one
                // group of slack; a group whose chunks are exhausted re-reads its current fragment)
                long g = (tap + 1 < K) ? ((long)c * K + tap + 1) : ((long)(c + KS) * K);
                if (KS > 1 && tap + 1 == K && c + KS >= nchunks) g = (long)c * K + tap;
                g *= 3 * 64;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int q = 0; q < 3; ++q) a_nxt[mi][q] = wp[mi][g + q * 64];
                if constexpr (kPipe) {
#pragma unroll
                    for (int i = 0; i < G::kNStage; ++i)
                        if (lt_of(i) == tap) stage_load_item(i, c + 1);
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the prefetch a whole tap ahead of its use
                if constexpr (kPipe) {
#pragma unroll
                    for (int i = 0; i < G::kNStage; ++i)
                        if (lt_of(i) + 2 == tap) stage_store_item(i, nxt);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    u32x4 bq[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        bq[q] = *reinterpret_cast<const u32x4 *>(cur + q * G::kPartBytes + (ni * 32 + tap * D) * kColBytes);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        constexpr int pa[6] = {2, 1, 0, 1, 0, 0};   // smallest products first
                        constexpr int pb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                        for (int t = 0; t < 6; ++t)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_cur[mi][pa[t]]),
                                                                                  __builtin_bit_cast(bf16x8, bq[pb[t]]),
                                                                                  acc[mi][ni], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int q = 0; q < 3; ++q) a_cur[mi][q] = a_nxt[mi][q];
            }
        }
        if constexpr (kPipe) {
#pragma unroll
            for (int i = 0; i < G::kNStage; ++i)
                if (lt_of(i) + 2 >= K) stage_store_item(i, nxt);
        } else {
#if TTSAMD_X3_PIPE == 0
            if (it + 1 < niter) stage_store(nxt);
#else
            stage_store(nxt);
#endif
        }
        __syncthreads();
    }

    if constexpr (KS > 1) {
        // partial tiles of groups 1..KS-1 -> LDS -> group 0 (fixed order), which alone runs the epilogue
        float *red = reinterpret_cast<float *>(xs3_all + (size_t)KS * G::kLdsBytes);
        constexpr int kTile = MI * NI * 16 * 64;            // floats of one wave's accumulators
        if (grp > 0) {
            float *dst = red + ((size_t)(grp - 1) * (WM * WN) + wave) * kTile + lane;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[((mi * NI + ni) * 16 + r) * 64] = acc[mi][ni][r];
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll 1
        for (int g = 1; g < KS; ++g) {
            const float *src = red + ((size_t)(g - 1) * (WM * WN) + wave) * kTile + lane;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] += src[((mi * NI + ni) * 16 + r) * 64];
        }
    }

#ifdef TTSAMD_PHASE_CLOCKS
    pc[2] = clock64();
#endif
    conv_epilogue<MODE, MI, NI, WM, WN>(acc, b, mb, t0, wm, wn, h, j, folded);
#ifdef TTSAMD_PHASE_CLOCKS
    pc[3] = clock64();
    __builtin_amdgcn_s_waitcnt(0);
    pc[4] = clock64();
    if (MODE == TTSAMD_CONV_NORMAL && a.y2 && tid == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == gridDim.z / 2) {
        long long *o = reinterpret_cast<long long *>(a.y2);
        for (int i = 0; i < 5; ++i) o[i] = pc[i] - pc[0];
        o[5] = wall_clock64() - prt0;
    }
#endif
}

template <int K, int D, int MI, int NI, int WM, int WN, int MODE, int KS = 1>
int conv1d_x3_launch_cfg(const ttsamd_conv1d_args &a, hipStream_t st)
{
    using G = ConvGeomX3<K, D, MI, NI, WM, WN>;
    auto kern = conv1d_x3_kernel<K, D, MI, NI, WM, WN, MODE, KS>;
    constexpr size_t kLds = (size_t)KS * G::kLdsBytes + (size_t)(KS - 1) * (WM * WN) * MI * NI * 16 * 64 * sizeof(float);
    static std::atomic<unsigned long long> lds_attr_done{0};   // per device, see ensure_dynamic_lds
    TTSAMD_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), (int)kLds, lds_attr_done));
    const int mtiles = (a.c_out + 31) / 32;
    const int mblocks = (mtiles + MI * WM - 1) / (MI * WM);
    const int nblocks = (a.t_out + G::kBN - 1) / G::kBN;
    hipLaunchKernelGGL(kern, dim3(nblocks, mblocks, a.batch), dim3(G::kThreads * KS), kLds, st, a);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

template <int K, int D, int MI, int MODE>
bool conv1d_x3s_launch(const ttsamd_conv1d_args &a, hipStream_t st, int *rc);   // conv_kernel_x3s.h
template <int K, int D, int MODE>
bool conv1d_x3o_launch(const ttsamd_conv1d_args &a, hipStream_t st, int *rc);   // conv_kernel_x3o.h
template <int K, int D, int MODE>
int conv1d_h2_launch_tiles(const ttsamd_conv1d_args &a, hipStream_t st);        // conv_kernel_h2.h
template <int K, int D, int MODE>
int conv1d_h2_launch_mid(const ttsamd_conv1d_args &a, hipStream_t st);          // conv_kernel_h2.h: 128 rows x 64 columns per block

template <int K, int D, int MODE>
int conv1d_x3_launch_tiles(const ttsamd_conv1d_args &a, hipStream_t st)
{
    const int mtiles = (a.c_out + 31) / 32;
    // Small grids (a B = 1 request: text encoder / duration predictor / flow layers at T = 257..770 launch 6-40 of the
    // default blocks on a 256-CU chip, and every block walks its serial K loop alone on its CU): 64-column tiles with ONE
    // 32x32 tile per wave — a quarter of the MFMA and staging work per k-step and 2-4x the blocks; with >= 8 channel chunks
    // four wave groups per block also split the K loop (fixed-order reduction through LDS: deterministic, but a different
    // summation order than the large-grid tiles — the usual fp32 reassociation, within every parity tolerance).  Only
    // instantiated where such launches occur: every NORMAL-mode conv (text side, flows, and the waveform decoder's 512/256-
    // channel stages of a single utterance) and the dilation-1 gate / res-skip / coupling convs of the flows.
    // (round 3: + the 1x1 affine-coupling convs of the Glow decoder — a single sentence launched THREE 64x256 blocks per flow
    // block there, 30 us each: 11 % of the Glow-TTS + HiFiGAN-v2 sentence)
    if constexpr (MODE == TTSAMD_CONV_SHUFFLE && K == 2) {
        // a single sentence's polyphase ConvTranspose launches (HiFiGAN-v2 ups[0]: 12 of the 128x128-class blocks): the
        // one-shot small-grid kernel, whole-tile epilogue (conv_kernel_x3o.h)
        const long blocks_default = (long)((a.t_out + 127) / 128) * ((mtiles + 3) / 4) * a.batch;
        if (g_conv_small_grid >= 4 && blocks_default <= kConvSmallGridBlocks) {
            int rc = TTSAMD_OK;
            if (conv1d_x3o_launch<K, D, MODE>(a, st, &rc)) return rc;
        }
        // a lone request's ups[0] (512 -> 256 x 8 rows at T = 770: 112 blocks of 128 x 128 on 256 CUs, 64 chunks each): the mid-size
        // tile of the NORMAL convs below (128 x 64 on eight waves).  TTSAMD_H2_MID_SHUFFLE=0: the large-grid tile (A/B switch)
        static const bool mid_shuffle = !(getenv("TTSAMD_H2_MID_SHUFFLE") && getenv("TTSAMD_H2_MID_SHUFFLE")[0] == '0');
        if (mid_shuffle && a.w_h2 && conv_h2_offsets_ok(a) && g_conv_small_grid && mtiles % 4 == 0 && blocks_default >= g_conv_h2_mid_min &&
            blocks_default <= kConvSmallGridBlocks)
            return conv1d_h2_launch_mid<K, D, MODE>(a, st);
    }
    constexpr bool affine = (MODE == TTSAMD_CONV_COUPLE_AFFINE || MODE == TTSAMD_CONV_COUPLE_AFFINE_FWD || MODE == TTSAMD_CONV_COUPLE_AFFINE_MIX);
    if constexpr (MODE == TTSAMD_CONV_NORMAL || (affine && K == 1) ||
                  (D == 1 && K <= 7 && (MODE == TTSAMD_CONV_GATE || MODE == TTSAMD_CONV_RES_SKIP || MODE == TTSAMD_CONV_COUPLE))) {
        const long tiles_n = (a.t_out + 127) / 128;
        const long blocks_default = tiles_n * ((mtiles + 3) / 4) * a.batch;      // 128x128-class blocks
        if constexpr (MODE == TTSAMD_CONV_NORMAL) {
            // Mid-size grids (one utterance's 256-channel decoder stage: 98 of the 128x128-class blocks): too few blocks for the
            // large-grid tile, long enough reductions that the K-split small-grid tiles pay six products per output — the
            // three-product kernel on 128-row x 64-column blocks (twice the blocks, half the serial chain of the 128x128 tile)
            if (a.w_h2 && conv_h2_offsets_ok(a) && g_conv_small_grid && mtiles % 4 == 0 && blocks_default >= g_conv_h2_mid_min && blocks_default <= kConvSmallGridBlocks)
                return conv1d_h2_launch_mid<K, D, MODE>(a, st);
        }
        if (g_conv_small_grid && blocks_default <= kConvSmallGridBlocks) {
            // mode 4 (default), first choice: the one-shot kernels of conv_kernel_x3o.h (every K slice its own wave, the whole
            // reduction in flight at once, the epilogue spread over four waves)
            if (g_conv_small_grid >= 4) {
                int rc = TTSAMD_OK;
                if (conv1d_x3o_launch<K, D, MODE>(a, st, &rc)) return rc;
            }
            // >= 8 channel chunks (c_in >= 128): four wave groups split the chunks of the block's K loop between them
            const bool ksplit = g_conv_small_grid > 1 && a.c_in >= 8 * kConvCK;
            if constexpr (TTSAMD_X3S_ALL || (D == 1 && K <= 5)) {
                // the looping small-grid kernels of conv_kernel_x3s.h
                if (ksplit && g_conv_small_grid >= 3) {
                    int rc = TTSAMD_OK;
                    if (conv1d_x3s_launch<K, D, 1, MODE>(a, st, &rc)) return rc;
                }
            }
            // (the paired-row modes pair inside a 32-row tile since round 4: they tile like every other mode)
            if (mtiles % 2 == 0) {                                                                 // 64 rows x 64 columns
                if (ksplit) return conv1d_x3_launch_cfg<K, D, 1, 1, 2, 2, MODE, 4>(a, st);
                return conv1d_x3_launch_cfg<K, D, 1, 1, 2, 2, MODE>(a, st);
            }
            return conv1d_x3_launch_cfg<K, D, 1, 1, 1, 2, MODE>(a, st);                            // 32 rows x 64 columns
        }
    }
    // 128x128 block as four 32x128 wave tiles: 3 weight loads (L2) + 12 LDS fragment reads per 24 MFMAs instead of
    // 6 + 6 — LDS has 4x the L1 bandwidth (scripts/ubench/x3_tiles.hip: 0.54 -> 0.56 of peak at the throttled clock,
    // 0.70 -> 0.79 on zero operands)
    // measured end to end (bench.py, two runs each): <2,2,2,2> 91.1 ms/step, <1,4,4,1> 87.3; 256-row blocks
    // (<2,4,4,1>) for the 2-tap polyphase ConvTranspose launches: +-0.  Round 2, same method: 64x128 wave tiles
    // (<2,4,2,1>, half the L1 bytes per MFMA, 20 spilled VGPRs) 79.5 -> 81.7 ms/step; 256-row blocks of 8 waves (<1,4,8,1>,
    // the 256-channel layers stage their tile once instead of twice) 77.5 -> 77.9; the conflict-free planar LDS image
    // (TTSAMD_X3_PLANAR) +-0 on the 128/256-row layers (SQ_LDS_BANK_CONFLICT 0, but LDS was not the limiter).
    // large grids with the two-part fp16 image present: three products per fp32 product instead of six (conv_kernel_h2.h)
    if (a.w_h2 && conv_h2_offsets_ok(a)) return conv1d_h2_launch_tiles<K, D, MODE>(a, st);
    if (mtiles % 4 == 0) return conv1d_x3_launch_cfg<K, D, TTSAMD_X3_CFG128, MODE>(a, st);
    // same bench, same box: <2,2,1,4> 90.6 ms/step, <1,4,2,2> 89.7; (<1,8,4,1> on the 128-row blocks: 93.4)
    if (mtiles % 2 == 0) return conv1d_x3_launch_cfg<K, D, TTSAMD_X3_CFG64, MODE>(a, st);
    return conv1d_x3_launch_cfg<K, D, 1, 2, 1, 4, MODE>(a, st);
}

}  // namespace ttsamd
