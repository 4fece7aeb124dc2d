// One ResBlock1 iteration of the HiFiGAN MRF as ONE kernel (TTS/vocoder/models/hifigan_generator.py:90-98):
//
//     mid = conv1(lrelu(x * mask), kernel K, dilation D) + bias1
//     y   = conv2(lrelu(mid * mask), kernel K, dilation 1) + bias2 + x   [+ accum] [/ div]
//
// Unfused, the pair is two launches and five tensor passes over HBM (x, mid written, mid read, x again as the residual,
// y); here `mid` never leaves the CU: conv1's accumulators are biased / masked / leaky-ReLU'd / split into the three
// bf16 parts in registers and written straight into the LDS image that conv2's B fragments are read from.  HBM sees x
// once (plus the halo overlap of neighbouring time tiles and the residual re-read of the tile's own columns, both L2
// hits) and y once: 5 passes -> 2.  Arithmetic is the split-bf16 scheme of conv_kernel_x3.h, product for product and in
// the same order, so the result is BITWISE what the two conv1d_x3 launches produce (tests/test_resblock_gpu.py).
//
// Block = all C channels (C = c_in = c_out in {32, 64, 128}) x kNM mid columns; waves are arranged WM x WN over
// (32-row m-tiles, 32-column n-tiles).  conv2 computes kNM columns too, of which kBN = kNM - (K-1) are valid outputs
// (the rest are the halo mid columns' worth of work: (K-1)/kNM of conv2's MFMAs, 4 % at K = 11, kNM = 256).
//
// LDS image (x tile, then — after a barrier — the mid tile in the same bytes): [part 3][chunk C/16][half 2][column][8 ch]
// bf16.  A B fragment (lane = column, half-wave = 8-channel half) is one ds_read_b128 at a 16-byte column stride inside
// one plane: the 16 lanes the LDS serves per cycle cover 16 different 16-byte slots -> conflict free (the
// [column][16 ch] image of conv_kernel_x3.h is 2-way conflicted on ds_read_b128's lane groups), and tap / dilation /
// n-tile / plane offsets are compile-time immediates.
#pragma once
#include "conv_kernel_x3.h"

#include <cstdlib>

namespace ttsamd {

using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

template <int K, int D, int C, int WM, int WN, int NI>
struct ResGeom {
    static constexpr int kThreads = 64 * WM * WN;
    static constexpr int kMI = C >= 32 * WM ? C / (32 * WM) : 1;   // m-tiles per wave (C = 16: one 32-row tile, half of it padding)
    static constexpr int kNCh = (C + 15) / 16;     // 16-channel chunks of the reduction
    static constexpr int kNM = 32 * NI * WN;       // mid columns computed (= conv2 columns computed)
    static constexpr int kBN = kNM - (K - 1);      // valid output columns per block
    static constexpr int kH2 = (K - 1) / 2;        // conv2 halo (dilation 1)
    static constexpr int kH1 = (K - 1) * D / 2;    // conv1 halo
    static constexpr int kXW = kNM + (K - 1) * D;  // x tile columns
    static constexpr int kXWm = kNM + (K - 1);     // mid tile columns (conv2 reads up to column kNM - 1 + K - 1)
    static constexpr int kPlaneX = kXW * 16;       // bytes of one [column][8 ch] plane of the x tile
    static constexpr int kPlaneM = kXWm * 16;
    static constexpr int kItems = kNCh * 2 * kXW;  // staged (chunk, half, column) items: 8 channels each
    static constexpr int kRounds = (kItems + kThreads - 1) / kThreads;
    static constexpr size_t kLdsBytes = (size_t)3 * kNCh * 2 * kPlaneX;
    static constexpr int kOcc = (NI == 1 && kThreads <= 256 && 4 * kLdsBytes <= 160 * 1024) ? 4     // small tiles: 4 blocks / CU
                                : (2 * kLdsBytes <= 160 * 1024 && kThreads <= 256) ? 2 : (kThreads >= 512 ? 2 : 1);
    static_assert((C % (32 * WM) == 0 || (C == 16 && WM == 1)) && kBN > 0, "bad tile");
};

// Main loop of one conv of the pair: acc[mi][ni] += sum over (chunk, tap) of the six split products.
//   wp[mi]  this wave's A stream (m-tile (wm*MI + mi)): [chunk][tap][part][64 lanes] x 16 bytes, read from L2/L1 one tap
//           ahead; a_cur holds tap 0 of chunk 0 on entry.
//   bbase   LDS address of this lane's fragment for (part 0, chunk 0, n-tile 0, tap 0): plane `half`, column wn*32*NI + j.
// Weight fragments are requested TWO taps ahead (a_cur: this tap, a_n1: the next, the third set is requested at the top of
// the tap): a wave of these kernels issues only 12 MFMAs per tap (MI = 1, NI = 2: 384 cycles), which covered an L1 hit but
// not the L2 round trip the fragments take whenever the streaming x tiles have pushed them out of L1 — the PMC counters of
// round 2 show it (matrix pipe busy 0.65-0.70 here against 0.78 in the unfused kernel, whose waves issue 24 MFMAs per tap).
template <int KK, int DD, int MI, int NI, int NCH, int PLANE>
__device__ __forceinline__ void res_conv_mainloop(f32x16 (&acc)[MI][NI], const u32x4 *const (&wp)[MI], u32x4 (&a_cur)[MI][3],
                                                  u32x4 (&a_n1)[MI][3], const unsigned char *bbase)
{
    u32x4 a_n2[MI][3];
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        const unsigned char *cb = bbase + c * (2 * PLANE);
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            const long g = ((long)c * KK + tap + 2) * (3 * 64);   // the packed image ends with two groups of slack
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int q = 0; q < 3; ++q) a_n2[mi][q] = wp[mi][g + q * 64];
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch two whole taps ahead of its use
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                u32x4 bq[3];
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    bq[q] = *reinterpret_cast<const u32x4 *>(cb + q * (NCH * 2 * PLANE) + (ni * 32 + tap * DD) * 16);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    constexpr int pa[6] = {2, 1, 0, 1, 0, 0};   // smallest products first (as conv1d_x3_kernel)
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
                for (int q = 0; q < 3; ++q) {
                    a_cur[mi][q] = a_n1[mi][q];
                    a_n1[mi][q] = a_n2[mi][q];
                }
        }
    }
}

#ifdef TTSAMD_PHASE_CLOCKS
// debug build only (scripts/res_phase.py): shader-clock stamps of one mid-grid block's wave 0 at the phase boundaries
static __device__ long long g_res_clk[16];     // one per translation unit (no relocatable device code): getters in resblock_k*.hip
#define TTSAMD_RES_CLOCK_GETTER(name)                                                                                    \
    extern "C" int name(long long *host_out16)                                                                           \
    {                                                                                                                    \
        return hipMemcpyFromSymbol(host_out16, HIP_SYMBOL(ttsamd::g_res_clk), 16 * sizeof(long long)) == hipSuccess ? 0 : -1; \
    }
#define RES_STAMP(i) do { if (stamp) { if ((i) == 0) g_res_clk[15] = wall_clock64(); g_res_clk[i] = clock64(); if ((i) == 8) g_res_clk[14] = wall_clock64(); } } while (0)
#else
#define RES_STAMP(i) do { } while (0)
#endif

typedef const ttsamd_resblock_args __attribute__((address_space(4))) *ResArgsKernargPtr;

// The block's work.  `a`: the pair's arguments (a kernel parameter of the caller), `ep`: the same struct inside the kernarg
// segment (the epilogue re-reads its arguments from there, see below), `tile`: (time tile, -, batch item) of this block.
template <int K, int D, int C, int WM, int WN, int NI>
__device__ __forceinline__ void resblock_pair_body(const ttsamd_resblock_args &a, ResArgsKernargPtr ep, const ConvTile tile)
{
#ifdef TTSAMD_PHASE_CLOCKS
    const bool stamp = threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.z == gridDim.z / 2;
#endif
    RES_STAMP(0);
    using G = ResGeom<K, D, C, WM, WN, NI>;
    constexpr int MI = G::kMI;
    constexpr int NCH = G::kNCh;
    extern __shared__ __attribute__((aligned(16))) unsigned char rs3[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int h = lane >> 5;
    const int j = lane & 31;
    const int b = tile.b;
    const int t0 = tile.nb * G::kBN;      // first output column of this block
    const int T = a.t;
    constexpr int kOob = kConvOob;

    // a.c may be smaller than the instantiation's C (HiFiGAN-v2's 16- and 8-channel stages run on the C = 32 kernel with
    // zero-padded weight images): every tensor is addressed with the REAL channel count, rows beyond it read as zeros
    // through the buffer range check and their stores are dropped by it
    const int creal = a.c;
    const long slab = (long)creal * T * 4;    // one item's [c, T] tensor (contiguous rows)
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)b * creal * T, slab);
    const __amdgpu_buffer_rsrc_t rmask = make_rsrc(a.mask ? a.mask + (long)b * T : nullptr, a.mask ? (long)T * 4 : 0);
    const bool has_mask = a.mask != nullptr;

    // ---- stage the x tile: columns [t0 - H2 - H1, +kXW) of all C channels, leaky-ReLU'd and split, into LDS ----------
    {
        const int tx0 = t0 - G::kH2 - G::kH1;
        const int row_bytes = T * 4;
        constexpr int kBatch = G::kRounds < 6 ? G::kRounds : 6;   // rounds whose loads are in flight together (8 dwords each)
#pragma unroll 1
        for (int r0 = 0; r0 < G::kRounds; r0 += kBatch) {
            float st[kBatch][8];
            float sm[kBatch];
#pragma unroll
            for (int rr = 0; rr < kBatch; ++rr) {
                const int e = tid + (r0 + rr) * G::kThreads;
                const int pl = e / G::kXW;                 // chunk * 2 + half
                const int col = e - pl * G::kXW;
                const int gt = tx0 + col;
                const bool ok = (e < G::kItems) && (gt >= 0) && (gt < T);
                const int off = ok ? (pl * 8 * row_bytes + gt * 4) : kOob;
#pragma unroll
                for (int i = 0; i < 8; ++i) st[rr][i] = ld_buf(rx, off == kOob ? kOob : off + i * row_bytes, 0);
                sm[rr] = has_mask ? ld_buf(rmask, ok ? gt * 4 : kOob, 0) : 1.f;
            }
#pragma unroll
            for (int rr = 0; rr < kBatch; ++rr) {
                const int e = tid + (r0 + rr) * G::kThreads;
                const int pl = e / G::kXW;
                const int col = e - pl * G::kXW;
                if (e < G::kItems) {
                    // two values per conversion instruction, parts stay packed (conv_split3x2: the same round-to-nearest-even
                    // conversions and exact residuals as conv_split3, bit for bit)
                    unsigned pw[3][4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        conv_split3x2(conv_lrelu(st[rr][2 * i] * sm[rr], a.slope), conv_lrelu(st[rr][2 * i + 1] * sm[rr], a.slope),
                                      pw[0][i], pw[1][i], pw[2][i]);
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        u32x4 w;
                        w.x = pw[q][0];
                        w.y = pw[q][1];
                        w.z = pw[q][2];
                        w.w = pw[q][3];
                        *reinterpret_cast<u32x4 *>(rs3 + (q * (NCH * 2) + pl) * G::kPlaneX + col * 16) = w;
                    }
                }
            }
        }
    }

    RES_STAMP(1);
    // ---- conv1 ------------------------------------------------------------------------------------------------------
    const u32x4 *wp1[MI], *wp2[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const long mtile = (long)wm * MI + mi;
        wp1[mi] = reinterpret_cast<const u32x4 *>(a.w1_split) + mtile * ((long)NCH * K * 3 * 64) + lane;
        wp2[mi] = reinterpret_cast<const u32x4 *>(a.w2_split) + mtile * ((long)NCH * K * 3 * 64) + lane;
    }
    u32x4 a_cur[MI][3], a_n1[MI][3];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            a_cur[mi][q] = wp1[mi][q * 64];
            a_n1[mi][q] = wp1[mi][(3 + q) * 64];
        }

    // conv2's accumulators start from the residual x — the tile's own columns, requested NOW: the lines were fetched for
    // the staging pass microseconds ago (L2 hits; requested after conv1 they had been evicted: PMC fetch 2.1x the tensor)
    // and their latency hides behind conv1's main loop
    f32x16 acc2[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row0 = (wm * MI + mi) * 32;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int o = (wn * NI + ni) * 32 + j;
            const int t = t0 + o;
            const int vo = (o < G::kBN && t < T) ? (4 * h * T + t) * 4 : kOob;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[mi][ni][r] = ld_buf(rx, vo, (row0 + (r & 3) + 8 * (r >> 2)) * T * 4);
        }
    }
    // operands of the mid epilogue, requested here so that their latency hides behind conv1's main loop
    float mk[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int t = t0 - G::kH2 + (wn * NI + ni) * 32 + j;      // time of this lane's mid column
        const bool ok = (t >= 0) && (t < T);                      // outside the tensor conv2 sees its zero padding
        mk[ni] = has_mask ? ld_buf(rmask, ok ? t * 4 : kOob, 0) : (ok ? 1.f : 0.f);
    }
    // biases through buffer resources: an absent bias is a zero-length resource (reads 0) — no branch per element
    const __amdgpu_buffer_rsrc_t rb1 = make_rsrc(a.bias1, a.bias1 ? creal * 4 : 0);
    const __amdgpu_buffer_rsrc_t rb2 = make_rsrc(a.bias2, a.bias2 ? creal * 4 : 0);
    float bia[MI][16];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) bia[mi][r] = ld_buf(rb1, 16 * h, ((wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2)) * 4);
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    __syncthreads();
    RES_STAMP(2);
    res_conv_mainloop<K, D, MI, NI, NCH, G::kPlaneX>(acc, wp1, a_cur, a_n1, rs3 + h * G::kPlaneX + (wn * (32 * NI) + j) * 16);
    RES_STAMP(3);

    // conv2's first weight fragments: requested before the mid epilogue so that their latency hides behind it
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            a_cur[mi][q] = wp2[mi][q * 64];
            a_n1[mi][q] = wp2[mi][(3 + q) * 64];
        }

    // ---- mid epilogue: (acc + bias1) * mask -> leaky ReLU -> 3-way split -> LDS (same bytes as the x tile) ------------
    {
        __syncthreads();                                               // every wave is done reading the x tile
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int col = (wn * NI + ni) * 32 + j;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    unsigned pw[3][2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float v0 = (acc[mi][ni][rg * 4 + 2 * i] + bia[mi][rg * 4 + 2 * i]) * mk[ni];
                        const float v1 = (acc[mi][ni][rg * 4 + 2 * i + 1] + bia[mi][rg * 4 + 2 * i + 1]) * mk[ni];
                        conv_split3x2(conv_lrelu(v0, a.slope), conv_lrelu(v1, a.slope), pw[0][i], pw[1][i], pw[2][i]);
                    }
                    // rows 8*rg + 4*h + i of m-tile (wm*MI + mi): chunk 2*mtile + rg/2, 8-channel half rg%2, channels 4h..4h+3
                    const int pl = (2 * (wm * MI + mi) + (rg >> 1)) * 2 + (rg & 1);
                    if (2 * (wm * MI + mi) + (rg >> 1) >= NCH) continue;     // C = 16: rows 16..31 of the tile are padding
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        u32x2 w;
                        w.x = pw[q][0];
                        w.y = pw[q][1];
                        *reinterpret_cast<u32x2 *>(rs3 + (q * (NCH * 2) + pl) * G::kPlaneM + col * 16 + h * 8) = w;
                    }
                }
            }
    }

    // ---- conv2 (its accumulators hold the residual x, requested before conv1) ---------------------------------------------
    float bia2[MI][16];      // output bias: requested ahead of conv2's main loop
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) bia2[mi][r] = ld_buf(rb2, 16 * h, ((wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2)) * 4);
    RES_STAMP(4);
    __syncthreads();                                                   // the mid tile is complete
    RES_STAMP(5);
    res_conv_mainloop<K, 1, MI, NI, NCH, G::kPlaneM>(acc2, wp2, a_cur, a_n1, rs3 + h * G::kPlaneM + (wn * (32 * NI) + j) * 16);
    RES_STAMP(6);

    // ---- output epilogue: + bias2 (+ accum) (/ div) -------------------------------------------------------------------
    {
        asm volatile("" : "+s"(ep) : : "memory");
        const float out_div = ep->out_div;
        const bool has_accum = ep->accum != nullptr;
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(ep->y + (long)b * creal * T, slab);
        const __amdgpu_buffer_rsrc_t racc = make_rsrc(has_accum ? ep->accum + (long)b * creal * T : nullptr, has_accum ? slab : 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row0 = (wm * MI + mi) * 32;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                __builtin_amdgcn_sched_barrier(0);
                const int o = (wn * NI + ni) * 32 + j;
                const int t = t0 + o;
                const int vo = (o < G::kBN && t < T) ? (4 * h * T + t) * 4 : kOob;
                // every optional pass behind ONE wave-uniform branch (inside the element loop hipcc evaluates the IEEE division
                // sequence for every element of every launch and selects afterwards); operations and their order are those of
                // the unfused conv epilogue
                float vout[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    vout[r] = acc2[mi][ni][r] + bia2[mi][r];
                    vout[r] += 0.f;              // (the unfused epilogue's absent-operand add: keeps -0.0 handling identical)
                }
                if (has_accum) {
                    float e2[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) e2[r] = ld_buf(racc, vo, (row0 + (r & 3) + 8 * (r >> 2)) * T * 4);
#pragma unroll
                    for (int r = 0; r < 16; ++r) vout[r] = e2[r] + vout[r];
                }
                if (out_div != 0.f) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) vout[r] = vout[r] / out_div;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) st_buf(ry, vout[r], vo, (row0 + (r & 3) + 8 * (r >> 2)) * T * 4);
            }
        }
    }
    RES_STAMP(7);
#ifdef TTSAMD_PHASE_CLOCKS
    __builtin_amdgcn_s_waitcnt(0);
#endif
    RES_STAMP(8);
}

template <int K, int D, int C, int WM, int WN, int NI>
__global__ __launch_bounds__(64 * WM * WN, (ResGeom<K, D, C, WM, WN, NI>::kOcc)) void resblock_pair_x3_kernel(const ttsamd_resblock_args a)
{
    resblock_pair_body<K, D, C, WM, WN, NI>(a, (ResArgsKernargPtr)__builtin_amdgcn_kernarg_segment_ptr(), conv_tile_of_block());
}

// The MRF's three branches (kernel sizes 3, 7, 11 — slot 0, 1, 2; a slot with x == NULL is absent) at one dilation as ONE launch:
// blockIdx.y selects the branch.  A single sentence runs 9 fused pairs per stage on three branch streams — 10-20 us kernels of
// 40-700 blocks whose cross-stream joins cost as much as the kernels (hipGraph replays them with ~10 us per cross-queue edge);
// grouped, a stage is three launches on one stream, each as long as its longest branch.  The grid is sized for the branch with
// the most time tiles (k = 11: the widest halo, the fewest valid columns per block); the others' surplus blocks return at once.
struct ResGroupArgs {
    ttsamd_resblock_args br[3];
};
template <int D, int C, int WM, int WN, int NI>
__global__ __launch_bounds__(64 * WM * WN, (ResGeom<11, D, C, WM, WN, NI>::kOcc)) void resblock_group_x3_kernel(const ResGroupArgs g)
{
    const ResArgsKernargPtr ep = (ResArgsKernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    const ConvTile tile{(int)blockIdx.x, 0, (int)blockIdx.z};
    const int br = blockIdx.y;
    if (br == 0) {
        if (g.br[0].x && tile.nb * ResGeom<3, D, C, WM, WN, NI>::kBN < g.br[0].t) resblock_pair_body<3, D, C, WM, WN, NI>(g.br[0], ep, tile);
    } else if (br == 1) {
        if (g.br[1].x && tile.nb * ResGeom<7, D, C, WM, WN, NI>::kBN < g.br[1].t) resblock_pair_body<7, D, C, WM, WN, NI>(g.br[1], ep + 1, tile);
    } else {
        if (g.br[2].x && tile.nb * ResGeom<11, D, C, WM, WN, NI>::kBN < g.br[2].t) resblock_pair_body<11, D, C, WM, WN, NI>(g.br[2], ep + 2, tile);
    }
}

template <int K, int D, int C, int WM, int WN, int NI>
int resblock_pair_launch_cfg(const ttsamd_resblock_args &a, hipStream_t st)
{
    using G = ResGeom<K, D, C, WM, WN, NI>;
    auto kern = resblock_pair_x3_kernel<K, D, C, WM, WN, NI>;
    static std::atomic<unsigned long long> lds_attr_done{0};
    TTSAMD_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), (int)G::kLdsBytes, lds_attr_done));
    const int nblocks = (a.t + G::kBN - 1) / G::kBN;
    hipLaunchKernelGGL(kern, dim3(nblocks, 1, a.batch), dim3(G::kThreads), G::kLdsBytes, st, a);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

// Tile per channel count (a.variant selects alternatives for A/B measurements; 0 = default):
//   C = 32 : 4 waves as 1x4, NI = 2 -> 256 mid columns, 59 KB LDS at K=11 D=5 (2 blocks / CU)
//   C = 64 : 4 waves as 2x2, NI = 2 -> 128 mid columns, 68 KB (2 blocks / CU);  variant 1: 8 waves 2x4, 256 columns, 117 KB
//            (measured at the benchmark shape, us per pair, 4-wave / 8-wave: k=3 988 / 1088, k=7 1760 / 1773, k=11 2686 / 2605:
//            k = 11 takes the 8-wave tile by default, variant 1 flips the choice)
//   C = 128: 8 waves as 4x2, NI = 2 -> 128 mid columns, 137 KB (1 block / CU)
template <int K, int D>
int resblock_pair_launch_kd(const ttsamd_resblock_args &a, hipStream_t st)
{
    if constexpr (K == 3) {
        // variant 2 (A/B): half-width tiles — 4 waves x 32 columns at C = 32, 2x2 waves x 32 columns at C = 64: a quarter /
        // half of the LDS, <= 128 VGPRs, 4 blocks per CU instead of 3 / 2 (more blocks in different phases per SIMD)
        if (a.variant == 2 && a.c == 32) return resblock_pair_launch_cfg<K, D, 32, 1, 4, 1>(a, st);
        if (a.variant == 2 && a.c == 64) return resblock_pair_launch_cfg<K, D, 64, 2, 2, 1>(a, st);
    }
    // Small grids (a single sentence through HiFiGAN-v2: the 64-channel stage is 2 700 columns = 11 of the default blocks, each
    // walking 2 x 44 k-steps alone on its CU): narrow tiles — a quarter / half of the columns per block, 4-5x the blocks.  Same
    // summation order per output: results do not depend on the tile width.
    const long cols = (long)a.t * a.batch;
    switch (a.c) {
        case 8:
        case 16:   // zero-padded weight images of the [32, 32, k] conv (the caller passes them); the reduction walks only the
                   // first 16-channel chunk (round 4: the 32-channel tile did twice the MFMA and staging work on zeros)
            // a sentence's 40-90 k columns: 128-column tiles (twice the blocks, half the chain per block: sentence -1 %, same box)
            if (a.variant == 0 && cols <= 512 * 236) return resblock_pair_launch_cfg<K, D, 16, 1, 4, 1>(a, st);
            return resblock_pair_launch_cfg<K, D, 16, 1, 4, 2>(a, st);
        case 32:
            if (a.variant == 0 && cols <= 128 * 236) return resblock_pair_launch_cfg<K, D, 32, 1, 4, 1>(a, st);
            return resblock_pair_launch_cfg<K, D, 32, 1, 4, 2>(a, st);
        case 64:
            if (a.variant == 0 && cols <= 64 * 118) return resblock_pair_launch_cfg<K, D, 64, 2, 2, 1>(a, st);
            // k = 11: the 8-wave / 256-column tile (4 % halo work instead of 8 %) wins by 3 %; k = 3, 7: the 4-wave tile
            if ((a.variant == 1) != (K == 11)) return resblock_pair_launch_cfg<K, D, 64, 2, 4, 2>(a, st);
            return resblock_pair_launch_cfg<K, D, 64, 2, 2, 2>(a, st);
        case 128: return resblock_pair_launch_cfg<K, D, 128, 4, 2, 2>(a, st);
    }
    set_error("resblock_pair: c = %d has no instantiation (8, 16, 32, 64, 128)", a.c);
    return TTSAMD_ERR_UNSUPPORTED;
}

// Grouped launches exist for the narrow small-grid tiles of resblock_pair_launch_kd (a single sentence).  One utterance's 64- /
// 32-channel stages (100-200 k columns, the default 4-wave tiles) were measured too: B = 1 request 4.74 -> 4.79 ms, not kept.
inline bool resblock_group_small(int c, long cols)
{
    return ((c == 8 || c == 16) && cols <= 512 * 236) || (c == 32 && cols <= 128 * 236) || (c == 64 && cols <= 64 * 118);
}

template <int D, int C, int WM, int WN, int NI>
int resblock_group_launch_cfg(const ResGroupArgs &g, int t, int batch, hipStream_t st)
{
    using G = ResGeom<11, D, C, WM, WN, NI>;      // the largest LDS image and the most blocks of the three
    auto kern = resblock_group_x3_kernel<D, C, WM, WN, NI>;
    static std::atomic<unsigned long long> lds_attr_done{0};
    TTSAMD_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), (int)G::kLdsBytes, lds_attr_done));
    const int nblocks = (t + G::kBN - 1) / G::kBN;
    hipLaunchKernelGGL(kern, dim3(nblocks, 3, batch), dim3(G::kThreads), G::kLdsBytes, st, g);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

template <int D>
int resblock_group_launch_d(const ResGroupArgs &g, int c, int t, int batch, hipStream_t st)
{
    switch (c) {
        case 8:
        case 16: return resblock_group_launch_cfg<D, 16, 1, 4, 1>(g, t, batch, st);
        case 32: return resblock_group_launch_cfg<D, 32, 1, 4, 1>(g, t, batch, st);
        case 64: return resblock_group_launch_cfg<D, 64, 2, 2, 1>(g, t, batch, st);
    }
    set_error("resblock_group: c = %d has no instantiation (8, 16, 32, 64)", c);
    return TTSAMD_ERR_UNSUPPORTED;
}

template <int K>
int resblock_pair_launch_k(const ttsamd_resblock_args &a, hipStream_t st)
{
    switch (a.dilation) {
        case 1: return resblock_pair_launch_kd<K, 1>(a, st);
        case 3: return resblock_pair_launch_kd<K, 3>(a, st);
        case 5: return resblock_pair_launch_kd<K, 5>(a, st);
    }
    set_error("resblock_pair: dilation %d has no instantiation (1, 3, 5)", a.dilation);
    return TTSAMD_ERR_UNSUPPORTED;
}

int resblock_pair_launch_k3(const ttsamd_resblock_args &a, hipStream_t st);
int resblock_pair_launch_k7(const ttsamd_resblock_args &a, hipStream_t st);
int resblock_pair_launch_k11(const ttsamd_resblock_args &a, hipStream_t st);
int resblock_group_launch_d1(const ResGroupArgs &g, int c, int t, int batch, hipStream_t st);
int resblock_group_launch_d3(const ResGroupArgs &g, int c, int t, int batch, hipStream_t st);
int resblock_group_launch_d5(const ResGroupArgs &g, int c, int t, int batch, hipStream_t st);

}  // namespace ttsamd
