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

template <int K, int D, int C, int WM, int WN, int NI>
__global__ __launch_bounds__(64 * WM * WN, (ResGeom<K, D, C, WM, WN, NI>::kOcc)) void resblock_pair_x3_kernel(const ttsamd_resblock_args a)
{
    const ConvTile tile = conv_tile_of_block();
#define RES_KERNARG_SLOT 0
#include "resblock_body.inc"
#undef RES_KERNARG_SLOT
}

// the same body as a function of the kernarg slot, for the grouped kernel below
template <int K, int D, int C, int WM, int WN, int NI, int SLOT>
__device__ __forceinline__ void resblock_pair_body(const ttsamd_resblock_args &a, const ConvTile tile)
{
#define RES_KERNARG_SLOT SLOT
#include "resblock_body.inc"
#undef RES_KERNARG_SLOT
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
    const ConvTile tile{(int)blockIdx.x, 0, (int)blockIdx.z};
    const int br = blockIdx.y;
    if (br == 0) {
        if (g.br[0].x && tile.nb * ResGeom<3, D, C, WM, WN, NI>::kBN < g.br[0].t) resblock_pair_body<3, D, C, WM, WN, NI, 0>(g.br[0], tile);
    } else if (br == 1) {
        if (g.br[1].x && tile.nb * ResGeom<7, D, C, WM, WN, NI>::kBN < g.br[1].t) resblock_pair_body<7, D, C, WM, WN, NI, 1>(g.br[1], tile);
    } else {
        if (g.br[2].x && tile.nb * ResGeom<11, D, C, WM, WN, NI>::kBN < g.br[2].t) resblock_pair_body<11, D, C, WM, WN, NI, 2>(g.br[2], tile);
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
inline bool resblock_group_small(int c, long cols);
template <int K, int D>
int resblock_pair_h2_launch_kd(const ttsamd_resblock_args &a, hipStream_t st);    // resblock_kernel_h2.h

template <int K, int D>
int resblock_pair_launch_kd(const ttsamd_resblock_args &a, hipStream_t st)
{
    // the default tiles with the two-part fp16 images present: three products per fp32 product (resblock_kernel_h2.h); the narrow
    // small-grid tiles below (a single sentence) keep the six-product kernel
    if (a.w1_h2 && a.w2_h2 && a.variant != 2 && !(a.variant == 0 && resblock_group_small(a.c, (long)a.t * a.batch)))
        return resblock_pair_h2_launch_kd<K, D>(a, st);
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
