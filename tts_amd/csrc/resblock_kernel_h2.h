// The fused ResBlock1 iteration of resblock_kernel_x3.h on the three-product arithmetic of conv_kernel_h2.h (two fp16 parts per
// fp32 operand, hi*hi in a main fp32 accumulator, the two cross products in a second one scaled by 2^-11 once):
//
//     mid = conv1(lrelu(x * mask), kernel K, dilation D) + bias1
//     y   = conv2(lrelu(mid * mask), kernel K, dilation 1) + bias2 + x   [+ accum] [/ div]
//
// Same block shape, LDS image order ([part 2][chunk][half][column][8 ch] fp16: two thirds of the split-bf16 image's bytes) and
// weight streams as the six-product kernel.  What the fp16 range adds: a block stages its WHOLE x tile at once and writes its
// whole mid tile at once, so each gets ONE power-of-two exponent per block, taken from the tile's largest magnitude:
//   * x tile: loads -> mask / leaky ReLU in registers -> wave maxima -> LDS slots -> barrier -> scale, split, write;
//   * mid tile: conv1's accumulators leave their units (activation exponent, row exponent), + bias, mask, leaky ReLU in
//     registers -> wave maxima -> slots -> the barrier that separates conv1's LDS reads from the mid writes anyway -> scale,
//     split, write.
// One barrier more than the six-product kernel (the x tile's).  The residual x is requested before conv1 as before, but kept in
// registers of its own and added in the output epilogue (the accumulators live in scaled units here).
// Results agree with two ttsamd_conv1d launches to fp32 rounding level, not bitwise: the unfused kernel scales per 16-channel
// chunk, this one per tile.
#pragma once
#include "conv_kernel_h2.h"
#include "resblock_kernel_x3.h"

namespace ttsamd {

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int K, int D, int C, int WM, int WN, int NI>
struct ResGeomH2 : ResGeom<K, D, C, WM, WN, NI> {
    using B = ResGeom<K, D, C, WM, WN, NI>;
    static constexpr int kCC = C < 32 ? 32 : C;                           // channel count of the (padded) weight images
    static constexpr size_t kImageBytes = (size_t)2 * B::kNCh * 2 * B::kPlaneX;
    static constexpr size_t kTabBytes = (size_t)4 * kCC * 4;              // [bias1 | row unscale 1 | bias2 | row unscale 2][row]
    static constexpr size_t kLdsBytes = kImageBytes + 2 * 8 * 4 + kTabBytes;   // + [x tile / mid tile][wave] maximum slots + tables
    // waves per SIMD the kernel is compiled for (= blocks per CU of a 4-wave block; an 8-wave block puts two waves on a SIMD):
    // as many as the LDS image allows, up to TTSAMD_PAIR_MAX_OCC — a block's life is mostly waiting (x tile from HBM, the two
    // epilogues, barriers), and only the other blocks of its CU fill the matrix pipe meanwhile
    // Measured (profiles/r06_pairs_occupancy_ab.txt): 2, 3 or 4 waves per SIMD are within 0.5 % of each other — the kernels are
    // issue-bound, not latency-bound — so the register budget goes to the residual instead: at <= 3 waves per SIMD (168 registers) the
    // tile's own x columns are requested together with the staging loads (the same cache lines: no second trip to HBM) and held
    // across both main loops.  With the request placed before conv2 or in the output epilogue (4 waves per SIMD, 128 registers) the
    // lines had left the L2 by then: FETCH_SIZE 2.0x the x tensor instead of 1.0x (profiles/r06_pair_traffic.txt).
#ifndef TTSAMD_PAIR_MAX_OCC
#define TTSAMD_PAIR_MAX_OCC 3
#endif
    static constexpr int kFit = (int)((160 * 1024) / kLdsBytes);         // blocks per CU by LDS
    static constexpr int kOccRaw = B::kThreads <= 256 ? kFit : 2 * kFit;
    static constexpr int kOcc = kOccRaw < 1 ? 1 : (kOccRaw > TTSAMD_PAIR_MAX_OCC ? TTSAMD_PAIR_MAX_OCC : kOccRaw);
    static constexpr bool kResEarly = kOcc <= 3;                          // 168+ registers: the residual is requested with the staging loads
    // waves that hold at least one item of the last (partly filled) staging round
    static constexpr int kLastWaves = (B::kItems - (B::kRounds - 1) * B::kThreads + 63) / 64;
    static_assert(WM * WN == 4 || WM * WN == 8, "the maximum slots are read four at a time");
};

// acc{m,x}[mi][ni] = sum over (chunk, tap) of the three products; weight fragments requested two taps ahead (a_cur: this tap,
// a_n1: the next), as in res_conv_mainloop.  The accumulators are OUTPUTS: the first products of chunk 0 / tap 0 take a zero C
// operand (an inline constant of the MFMA) instead of 64 v_mov of zero-initialisation per conv — on this chip a VALU instruction
// costs the SIMD the issue slots of an eighth of an MFMA (matrix-pipe busy + 4 cycles per other VALU instruction add up
// to the kernel's cycles).
template <bool FIRST, int KK, int DD, int MI, int NI, int NCH, int PLANE>
__device__ __forceinline__ void res_conv_chunk_h2(f32x16 (&accm)[MI][NI], f32x16 (&accx)[MI][NI], const u32x4 *const (&wp)[MI], u32x4 (&a_cur)[MI][2],
                                                  u32x4 (&a_n1)[MI][2], const unsigned char *cb, int c)
{
    u32x4 a_n2[MI][2];
    constexpr f32x16 kZero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) {
        const long g = ((long)c * KK + tap + 2) * (2 * 64);   // the packed image ends with two groups of slack
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int q = 0; q < 2; ++q) a_n2[mi][q] = wp[mi][g + q * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            u32x4 bq[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) bq[q] = *reinterpret_cast<const u32x4 *>(cb + q * (NCH * 2 * PLANE) + (ni * 32 + tap * DD) * 16);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const bool zero = FIRST && tap == 0;
                accx[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_cur[mi][1]), __builtin_bit_cast(f16x8, bq[0]), zero ? kZero : accx[mi][ni], 0, 0, 0);
                accx[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_cur[mi][0]), __builtin_bit_cast(f16x8, bq[1]), accx[mi][ni], 0, 0, 0);
                accm[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_cur[mi][0]), __builtin_bit_cast(f16x8, bq[0]), zero ? kZero : accm[mi][ni], 0, 0, 0);
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                a_cur[mi][q] = a_n1[mi][q];
                a_n1[mi][q] = a_n2[mi][q];
            }
    }
}

template <int KK, int DD, int MI, int NI, int NCH, int PLANE>
__device__ __forceinline__ void res_conv_mainloop_h2(f32x16 (&accm)[MI][NI], f32x16 (&accx)[MI][NI], const u32x4 *const (&wp)[MI],
                                                     u32x4 (&a_cur)[MI][2], u32x4 (&a_n1)[MI][2], const unsigned char *bbase)
{
    res_conv_chunk_h2<true, KK, DD, MI, NI, NCH, PLANE>(accm, accx, wp, a_cur, a_n1, bbase, 0);
#pragma unroll 1
    for (int c = 1; c < NCH; ++c) res_conv_chunk_h2<false, KK, DD, MI, NI, NCH, PLANE>(accm, accx, wp, a_cur, a_n1, bbase + c * (2 * PLANE), c);
}

template <int K, int D, int C, int WM, int WN, int NI>
__global__ __launch_bounds__(64 * WM * WN, (ResGeomH2<K, D, C, WM, WN, NI>::kOcc)) void resblock_pair_h2_kernel(const ttsamd_resblock_args a)
{
    using G = ResGeomH2<K, D, C, WM, WN, NI>;
    constexpr int MI = G::kMI;
    constexpr int NCH = G::kNCh;
    constexpr int NW = WM * WN;
    constexpr int CC = G::kCC;
    extern __shared__ __attribute__((aligned(16))) unsigned char rh2[];
    unsigned *const slots = reinterpret_cast<unsigned *>(rh2 + G::kImageBytes);
    float *const tabs = reinterpret_cast<float *>(rh2 + G::kImageBytes + 2 * 8 * 4);

    const ConvTile tile = conv_tile_of_block();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int h = lane >> 5;
    const int j = lane & 31;
    const int b = tile.b;
    const int t0 = tile.nb * G::kBN;
    const int T = a.t;
    constexpr int kOob = kConvOob;

    const int creal = a.c;                    // may be smaller than C (8 / 16 channels on the padded tile): real rows only
    const long slab = (long)creal * T * 4;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)b * creal * T, slab);
    const __amdgpu_buffer_rsrc_t rmask = make_rsrc(a.mask ? a.mask + (long)b * T : nullptr, a.mask ? (long)T * 4 : 0);
    const bool has_mask = a.mask != nullptr;

    auto block_max = [&](int which) -> unsigned {      // after a barrier
        const u32x4 s0 = *reinterpret_cast<const u32x4 *>(slots + which * 8);
        unsigned m = max(max(s0.x, s0.y), max(s0.z, s0.w));
        if constexpr (NW > 4) {
            const u32x4 s1 = *reinterpret_cast<const u32x4 *>(slots + which * 8 + 4);
            m = max(m, max(max(s1.x, s1.y), max(s1.z, s1.w)));
        }
        return (unsigned)__builtin_amdgcn_readfirstlane((int)m);
    };

    // the residual x — the tile's own columns, added in the output epilogue.  Requested right behind the staging loads (same lines)
    // where the register budget has room for it (<= 3 waves per SIMD), in the output epilogue at 4 waves per SIMD
    f32x16 resv[MI][G::kResEarly ? NI : 1];
    auto request_residual = [&](int mi, int ni, f32x16 &dst) {
        const int row0 = (wm * MI + mi) * 32;
        const int o = (wn * NI + ni) * 32 + j;
        const int t = t0 + o;
        const int vo = (o < G::kBN && t < T) ? (4 * h * T + t) * 4 : kOob;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r] = ld_buf(rx, vo, (row0 + (r & 3) + 8 * (r >> 2)) * T * 4);
    };

    // ---- stage the x tile: columns [t0 - H2 - H1, +kXW) of all C channels: mask, leaky ReLU, block exponent, split, LDS ------
    int e_x;
    {
        const int tx0 = t0 - G::kH2 - G::kH1;
        const int row_bytes = T * 4;
        // the last staging round is partly filled: waves without an item in it skip its loads, arithmetic and stores
        const bool last_round = wave < G::kLastWaves;
        float st[G::kRounds][8];
#pragma unroll
        for (int rr = 0; rr < G::kRounds; ++rr) {
            if (rr == G::kRounds - 1 && G::kLastWaves < NW && !last_round) {
#pragma unroll
                for (int i = 0; i < 8; ++i) st[rr][i] = 0.f;
                continue;
            }
            const int e = tid + rr * G::kThreads;
            const int pl = e / G::kXW;                 // chunk * 2 + half
            const int col = e - pl * G::kXW;
            const int gt = tx0 + col;
            const bool ok = (e < G::kItems) && (gt >= 0) && (gt < T);
            const int off = ok ? (pl * 8 * row_bytes + gt * 4) : kOob;
            float sm;
            // the row offset rides in the load's SCALAR offset: no VALU instruction per load.  The hardware range check of a raw buffer
            // access on gfx950 covers voffset + soffset (measured: scripts/ubench/soffset_range.hip), so rows beyond an 8- / 16-channel
            // tensor read as zeros and an invalid lane's kOob + i * row_bytes stays out of range (a slab is < 2 GiB: no wrap below 2^32)
#pragma unroll
            for (int i = 0; i < 8; ++i) st[rr][i] = ld_buf(rx, off, i * row_bytes);
            if (has_mask) {                         // block-uniform: an unmasked call pays no multiply per value
                sm = ld_buf(rmask, ok ? gt * 4 : kOob, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) st[rr][i] *= sm;
            }
        }
        // the epilogues' per-row operands (bias, 2^-e_row of the weight image) go through a small LDS table instead of living
        // in 32 registers across both main loops: written here, read after the barriers below
        if (tid < CC) {
            const __amdgpu_buffer_rsrc_t rb1 = make_rsrc(a.bias1, a.bias1 ? creal * 4 : 0);
            const __amdgpu_buffer_rsrc_t rb2 = make_rsrc(a.bias2, a.bias2 ? creal * 4 : 0);
            const float *const tab1 = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(a.w1_h2) + conv_h2_table_offset(CC, CC, K) + sizeof(H2RowTable));
            const float *const tab2 = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(a.w2_h2) + conv_h2_table_offset(CC, CC, K) + sizeof(H2RowTable));
            tabs[tid] = ld_buf(rb1, tid * 4, 0);
            tabs[CC + tid] = tab1[2 * tid + 1];
            tabs[2 * CC + tid] = ld_buf(rb2, tid * 4, 0);
            tabs[3 * CC + tid] = tab2[2 * tid + 1];
        }
        if constexpr (G::kResEarly) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) request_residual(mi, ni, resv[mi][ni]);
        }
        float m = 0.f;
#pragma unroll
        for (int rr = 0; rr < G::kRounds; ++rr) {
            if (rr == G::kRounds - 1 && G::kLastWaves < NW && !last_round) continue;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                st[rr][i] = conv_lrelu(st[rr][i], a.slope);
                m = __builtin_fmaxf(m, __builtin_fabsf(st[rr][i]));
            }
        }
        slots[wave] = wave_max_u32(__builtin_bit_cast(unsigned, m));
        __syncthreads();
        e_x = h2_exp_for(block_max(0));
        const float sx = pow2f(e_x);
#pragma unroll
        for (int rr = 0; rr < G::kRounds; ++rr) {
            if (rr == G::kRounds - 1 && G::kLastWaves < NW && !last_round) continue;
            const int e = tid + rr * G::kThreads;
            const int pl = e / G::kXW;
            const int col = e - pl * G::kXW;
            if (e < G::kItems) {
                unsigned pw[2][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) conv_split2x2(st[rr][2 * i] * sx, st[rr][2 * i + 1] * sx, pw[0][i], pw[1][i]);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    u32x4 w;
                    w.x = pw[q][0];
                    w.y = pw[q][1];
                    w.z = pw[q][2];
                    w.w = pw[q][3];
                    *reinterpret_cast<u32x4 *>(rh2 + (q * (NCH * 2) + pl) * G::kPlaneX + col * 16) = w;
                }
            }
        }
    }

    // ---- conv1 ------------------------------------------------------------------------------------------------------
    const u32x4 *wp1[MI], *wp2[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const long mtile = (long)wm * MI + mi;
        wp1[mi] = reinterpret_cast<const u32x4 *>(a.w1_h2) + mtile * ((long)NCH * K * 2 * 64) + lane;
        wp2[mi] = reinterpret_cast<const u32x4 *>(a.w2_h2) + mtile * ((long)NCH * K * 2 * 64) + lane;
    }
    u32x4 a_cur[MI][2], a_n1[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            a_cur[mi][q] = wp1[mi][q * 64];
            a_n1[mi][q] = wp1[mi][(2 + q) * 64];
        }
    // a lane's 16 rows of a table are four groups of four consecutive rows (row0 + 8 rg + 4 h + 0..3): one ds_read_b128 per group,
    // read where it is used (no table registers live across the main loops)
    auto table_row4 = [&](int which, int mi, int rg) -> f32x4 {
        return *reinterpret_cast<const f32x4 *>(tabs + which * CC + (wm * MI + mi) * 32 + 4 * h + 8 * rg);
    };
    f32x16 accm[MI][NI], accx[MI][NI];          // outputs of the main loops (their first products start from a zero C operand)
    __syncthreads();
    res_conv_mainloop_h2<K, D, MI, NI, NCH, G::kPlaneX>(accm, accx, wp1, a_cur, a_n1, rh2 + h * G::kPlaneX + (wn * (32 * NI) + j) * 16);

    // conv2's first weight fragments: requested before the mid epilogue so that their latency hides behind it
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            a_cur[mi][q] = wp2[mi][q * 64];
            a_n1[mi][q] = wp2[mi][(2 + q) * 64];
        }

    // ---- mid epilogue: leave the scaled units, + bias1, mask, leaky ReLU (in registers), tile exponent, split -> LDS -------------
    int e_m;
    {
        float mk[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int t = t0 - G::kH2 + (wn * NI + ni) * 32 + j;      // time of this lane's mid column
            const bool ok = (t >= 0) && (t < T);                      // outside the tensor conv2 sees its zero padding
            mk[ni] = has_mask ? ld_buf(rmask, ok ? t * 4 : kOob, 0) : (ok ? 1.f : 0.f);
        }
        const float usx = pow2f(-e_x);
        // the mask pass runs only where a factor can differ from 1: a masked call, or a block whose mid columns reach outside the tensor
        // (block-uniform branch around the whole pass: interior blocks of an unmasked call pay no multiply per value)
        const bool need_mask = has_mask || (t0 - G::kH2 < 0) || (t0 - G::kH2 + G::kNM > T);
        float m = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 bia = table_row4(0, mi, rg), ru = table_row4(1, mi, rg);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * rg + i;
                        // (x 2^-11 is exact: same bits as mul, add; x ru, a power of two, is exact too, so the fma below rounds what
                        // `(. * ru) + bias` rounded — one instruction less per value)
                        accm[mi][ni][r] = __builtin_fmaf(__builtin_fmaf(accx[mi][ni][r], 1.f / 2048.f, accm[mi][ni][r]) * usx, ru[i], bia[i]);
                    }
            }
        if (need_mask) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accm[mi][ni][r] *= mk[ni];
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = conv_lrelu(accm[mi][ni][r], a.slope);
                    accm[mi][ni][r] = v;
                    m = __builtin_fmaxf(m, __builtin_fabsf(v));
                }
        slots[8 + wave] = wave_max_u32(__builtin_bit_cast(unsigned, m));
        __syncthreads();                                               // every wave is done reading the x tile; the maxima are in
        e_m = h2_exp_for(block_max(1));
        const float sm = pow2f(e_m);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int col = (wn * NI + ni) * 32 + j;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    unsigned pw[2][2];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        conv_split2x2(accm[mi][ni][rg * 4 + 2 * i] * sm, accm[mi][ni][rg * 4 + 2 * i + 1] * sm, pw[0][i], pw[1][i]);
                    // rows 8*rg + 4*h + i of m-tile (wm*MI + mi): chunk 2*mtile + rg/2, 8-channel half rg%2, channels 4h..4h+3
                    const int pl = (2 * (wm * MI + mi) + (rg >> 1)) * 2 + (rg & 1);
                    if (2 * (wm * MI + mi) + (rg >> 1) >= NCH) continue;     // C = 16: rows 16..31 of the tile are padding
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        u32x2 w;
                        w.x = pw[q][0];
                        w.y = pw[q][1];
                        *reinterpret_cast<u32x2 *>(rh2 + (q * (NCH * 2) + pl) * G::kPlaneM + col * 16 + h * 8) = w;
                    }
                }
            }
    }

    // ---- conv2 ----------------------------------------------------------------------------------------------------------
    __syncthreads();                                                   // the mid tile is complete
    res_conv_mainloop_h2<K, 1, MI, NI, NCH, G::kPlaneM>(accm, accx, wp2, a_cur, a_n1, rh2 + h * G::kPlaneM + (wn * (32 * NI) + j) * 16);

    // ---- output epilogue: leave the scaled units, + bias2, + x (+ accum) (/ div) ---------------------------------------------
    {
        ResArgsKernargPtr ep = (ResArgsKernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ep) : : "memory");
        const float out_div = ep->out_div;
        const bool has_accum = ep->accum != nullptr;
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(ep->y + (long)b * creal * T, slab);
        const __amdgpu_buffer_rsrc_t racc = make_rsrc(has_accum ? ep->accum + (long)b * creal * T : nullptr, has_accum ? slab : 0);
        const float usm = pow2f(-e_m);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row0 = (wm * MI + mi) * 32;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                __builtin_amdgcn_sched_barrier(0);
                const int o = (wn * NI + ni) * 32 + j;
                const int t = t0 + o;
                const int vo = (o < G::kBN && t < T) ? (4 * h * T + t) * 4 : kOob;
                f32x16 &rv = resv[mi][G::kResEarly ? ni : 0];
                if constexpr (!G::kResEarly) request_residual(mi, ni, rv);     // 4 waves per SIMD: no room to hold it across conv2
                float e2[16];
                if (has_accum) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) e2[r] = ld_buf(racc, vo, (row0 + (r & 3) + 8 * (r >> 2)) * T * 4);
                }
                float vout[16];
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const f32x4 bia = table_row4(2, mi, rg), ru = table_row4(3, mi, rg);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * rg + i;
                        vout[r] = __builtin_fmaf(__builtin_fmaf(accx[mi][ni][r], 1.f / 2048.f, accm[mi][ni][r]) * usm, ru[i], bia[i]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) vout[r] += rv[r];
                if (has_accum) {
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
}

template <int K, int D, int C, int WM, int WN, int NI>
int resblock_pair_h2_launch_cfg(const ttsamd_resblock_args &a, hipStream_t st)
{
    using G = ResGeomH2<K, D, C, WM, WN, NI>;
    auto kern = resblock_pair_h2_kernel<K, D, C, WM, WN, NI>;
    static std::atomic<unsigned long long> lds_attr_done{0};
    TTSAMD_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), (int)G::kLdsBytes, lds_attr_done));
    const int nblocks = (a.t + G::kBN - 1) / G::kBN;
    hipLaunchKernelGGL(kern, dim3(nblocks, 1, a.batch), dim3(G::kThreads), G::kLdsBytes, st, a);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

// the default (large-grid) tiles of resblock_pair_launch_kd; the narrow small-grid tiles keep the six-product kernel
template <int K, int D>
int resblock_pair_h2_launch_kd(const ttsamd_resblock_args &a, hipStream_t st)
{
    switch (a.c) {
        case 8:
        case 16: return resblock_pair_h2_launch_cfg<K, D, 16, 1, 4, 2>(a, st);
        case 32:
            // variant 2: four n-tiles per wave (512 mid columns per block): half the weight-fragment traffic per MFMA (A/B only)
            if (a.variant == 2) return resblock_pair_h2_launch_cfg<K, D, 32, 1, 4, 4>(a, st);
            return resblock_pair_h2_launch_cfg<K, D, 32, 1, 4, 2>(a, st);
        case 64:
            // (one n-tile per wave — half the columns, three blocks per CU — measured 1.14-1.22x SLOWER at 64 and 32 channels)
            // the 4-wave / 128-column tile for every kernel size (two blocks per CU: one block's staging and epilogue phases run
            // under the other's MFMAs); on six products k = 11 preferred the 8-wave / 256-column tile (4 % halo work instead of
            // 8 %), on three it is 1.5-2 % slower (scripts/h2_variants_ab.py); variant 1 selects it for A/B
            if (a.variant == 1) return resblock_pair_h2_launch_cfg<K, D, 64, 2, 4, 2>(a, st);
            if (a.variant == 2) return resblock_pair_h2_launch_cfg<K, D, 64, 2, 2, 4>(a, st);      // 256 mid columns, four n-tiles per wave (A/B only)
            return resblock_pair_h2_launch_cfg<K, D, 64, 2, 2, 2>(a, st);
        case 128:
            // 4 waves x 64 mid columns: two blocks per CU (one block's staging / epilogue phases run under the other's MFMAs) instead
            // of one 8-wave / 128-column block — 0.92-0.98 of its time at k = 3 and 7, bitwise the same results
            // (scripts/h2_variants_ab.py); variant 1 selects the 8-wave tile for A/B
            if (a.variant == 1) return resblock_pair_h2_launch_cfg<K, D, 128, 4, 2, 2>(a, st);
            return resblock_pair_h2_launch_cfg<K, D, 128, 4, 1, 2>(a, st);
        case 256:
            // three-product arithmetic only (round 6): 8 waves x 32 rows, 64 mid columns, 76-117 KB of LDS, one block per CU.  The
            // unfused 256-channel convs stage every x tile once per 128 output rows and per conv; here it is staged once per pair.
            return resblock_pair_h2_launch_cfg<K, D, 256, 8, 1, 2>(a, st);
    }
    set_error("resblock_pair: c = %d has no instantiation (8, 16, 32, 64, 128, 256)", a.c);
    return TTSAMD_ERR_UNSUPPORTED;
}

}  // namespace ttsamd
