// Conv1d as implicit GEMM on the fp16 MFMA of gfx950 (v_mfma_f32_32x32x16_f16) with both fp32 operands split into TWO fp16
// parts and THREE products per fp32 product (conv_kernel_x3.h: three bf16 parts, six products):
//
//     x = hi + lo * 2^-11,   hi = fp16(x),   lo = fp16((x - hi) * 2^11)           (round to nearest even, residual exact)
//     w x  ~=  hi_w hi_x  +  2^-11 (hi_w lo_x + lo_w hi_x)                          (dropped: 2^-22 lo_w lo_x <= 2^-24 |w x|)
//
// `hi_w hi_x` goes into the main fp32 accumulator, the two cross products into a SECOND fp32 accumulator that is scaled by 2^-11
// once, in the epilogue.  The low part carries its own exponent (it is normal whenever the high part is), so a value keeps
// 11 + 11 significand bits and the sign of lo: 2^-23 relative representation error, about one fp32 rounding per operand.
// Measured against an fp64 conv on the adversarial operands of tests/test_conv_gpu.py: 1.1-2.1e-7 of sum|w x| (six bf16 products
// 0.4-1.1e-7, torch's fp32 CPU conv 0.5-1.4e-7, the exact fp32-input MFMA path 5.3e-7) — fp32-class, at half the matrix-pipe
// work of the six-product scheme (profiles/r05_h2_experiment.txt: 1.5-1.7x on the k = 7 / 11 convs at socket power).
//
// fp16 has 5 exponent bits, so both operands are brought into its range by exact powers of two:
//   * weights: one exponent per packed output row, chosen at pack time (row maximum in [2^13, 2^14)), stored with the image and
//     undone in the epilogue (ttsamd_conv1d_pack_weights_h2);
//   * activations: one exponent per (block, 16-channel chunk tile), derived IN the kernel from the tile's largest magnitude
//     (after mask / activation): the waves' maxima meet in LDS at the barrier the chunk pipeline has anyway.  The block keeps a
//     RUNNING exponent e (chunk maximum scaled into [2^12, 2^13) when it is set) and changes it only when a later chunk would
//     reach 2^15: then — rarely: chunk maxima of one tensor differ by a few binades — the accumulators are rescaled by the exact
//     power of two.  A value keeps full precision down to 2^-27 of its chunk's maximum and degrades gradually below that
//     (absolute error <= 2^-48 of the maximum): fp16's normal range plus the separately scaled low part.
// fp16 denormal operands are exact on this matrix pipe (measured: scripts/ubench/h2_bench.py).
//
// Everything else is conv1d_x3_kernel's design: A fragments straight from L2 in [m-tile][chunk][tap][part][64 lanes][8 fp16]
// order, prefetched a tap ahead; the B tile of a chunk staged once into the planar [part][half][column][8 ch] LDS image (one
// conflict-free ds_read_b128 per fragment, tap / dilation as immediates), double buffered; the fused epilogues of conv_kernel.h.
// Staging pipeline (one barrier per chunk, one set of staging registers): in iteration c the block converts chunk c+1 (loaded
// during iteration c-1, its maximum known since the last barrier), then requests chunk c+2 into the same registers, and takes
// that chunk's maximum just before the barrier.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "conv_kernel_x3.h"

namespace ttsamd {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2v = __attribute__((ext_vector_type(2))) _Float16;
using f32x2v = __attribute__((ext_vector_type(2))) float;

constexpr int kH2TargetExp = 12;                // a chunk maximum is scaled into [2^12, 2^13) when the running exponent is set
constexpr int kH2LimitExp = 15;                 // ... and the exponent is renewed when a chunk maximum would reach 2^15
constexpr int kH2FoldMaxExp = 60;               // residual folded into the accumulators only while activation + row exponent <= 60

template <int K, int D, int MI, int NI, int WM, int WN>
struct ConvGeomH2 {
    static constexpr int kThreads = 64 * WM * WN;
    static constexpr int kBM = 32 * MI * WM;
    static constexpr int kBN = 32 * NI * WN;
    static constexpr int kHalo = (K - 1) * D;
    static constexpr int kXW = kBN + kHalo;
    static constexpr int kXWp = kXW + 1;                     // + one dump column per plane (idle lanes of the last staging round)
    static constexpr int kPartBytes = kXWp * 32;             // [half][column][8 ch] fp16
    static constexpr int kBufBytes = 2 * kPartBytes;
    static constexpr int kItems = 2 * kXW;                      // (half, column) items of 8 channels
    // Staging work is dealt out evenly: kFull rounds in which every thread stages one 8-channel item, and the kRem items left
    // over (the halo columns: 4 of 260 items at k = 3 on the 128-column tile) as SINGLE elements, kSingles per thread — before
    // round 6 they cost every thread a whole extra 8-channel round (16 staged values per thread and chunk instead of 9).
    static constexpr int kFull = kItems / kThreads;
    static constexpr int kRem = kItems - kFull * kThreads;
    static constexpr int kSingles = (kRem * 8 + kThreads - 1) / kThreads;
    static constexpr int kNStage = kFull + (kSingles > 0 ? 1 : 0);   // staging steps (K >= 7 spreads them over the taps)
    // K < 7: too few MFMAs per chunk to cover an HBM round trip with one chunk of lead -> two register sets, requests two chunks ahead
#ifndef TTSAMD_H2_DEEP
#define TTSAMD_H2_DEEP 1
#endif
    static constexpr int kSets = (K < 7 && TTSAMD_H2_DEEP) ? 2 : 1;
    static constexpr int kSlotBytes = 2 * 8 * 4;             // [chunk parity][wave] largest magnitude (bit pattern)
    static constexpr size_t kLdsBytes = (size_t)2 * kBufBytes + kSlotBytes;
    static_assert(WM * WN == 4 || WM * WN == 8, "the maximum slots are read four at a time");
};

// two values -> packed high parts and packed (2^11-scaled) low parts.
// lo = fp16((x - hi) * 2^11) as ONE mixed-precision fma per value, fp16(fma(hi, -2^11, x * 2^11)), reading hi straight from its
// packed fp16 half: v_pk_mul + v_cvt_pk + v_fma_mixlo + v_fma_mixhi instead of v_cvt_pk + two v_cvt_f32_f16 + v_pk_add + v_pk_mul +
// v_cvt_pk (hipcc keeps the convert-back form, and turns a C-level fma pair into v_pk_fma_f32 + the same conversions).  x * 2^11,
// hi * 2^11 and their difference are exact, so both forms round once, to fp16, at the end: bitwise equal on 4 M random pairs over
// fp16's normal, denormal and underflow range (scripts/ubench/split_mix.hip).  TTSAMD_SPLIT_MIX=0 builds the convert-back form.
#ifndef TTSAMD_SPLIT_MIX
#define TTSAMD_SPLIT_MIX 1
#endif
__device__ __forceinline__ void conv_split2x2(float x0, float x1, unsigned &whi, unsigned &wlo)
{
    const f32x2v v = {x0, x1};
    const f16x2v hi = __builtin_convertvector(v, f16x2v);
    whi = __builtin_bit_cast(unsigned, hi);
#if TTSAMD_SPLIT_MIX
    const f32x2v v2k = v * 2048.f;
    const float m = -2048.f;
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(whi), "s"(m), "v"(v2k[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(whi), "s"(m), "v"(v2k[1]));
    wlo = lo;
#else
    const f32x2v hf = __builtin_convertvector(hi, f32x2v);
    const f32x2v r = (v - hf) * 2048.f;                                        // both steps exact
    wlo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2v));
#endif
}

// largest value of a wave (unsigned compare: magnitudes' bit patterns order like the magnitudes), wave-uniform result
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));   // row_half_mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));   // row_mirror: every lane = its row's maximum
    const unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}

__device__ __forceinline__ float pow2f(int e) { return __builtin_bit_cast(float, (unsigned)(e + 127) << 23); }   // e in [-126, 127]

// exponent that brings a chunk whose largest magnitude has bit pattern m into [2^12, 2^13) (clamped to what one fp32 factor holds)
__device__ __forceinline__ int h2_exp_for(unsigned m)
{
    const int e = kH2TargetExp + 127 - (int)(m >> 23);
    return e > 126 ? 126 : (e < -126 ? -126 : e);
}

template <int K, int D, int MI, int NI, int WM, int WN, int MODE>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv1d_h2_kernel(const ttsamd_conv1d_args a)
{
    using G = ConvGeomH2<K, D, MI, NI, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char xh2[];   // [2][2 parts][2 halves][XWp][8 ch] fp16, then the maximum slots
    unsigned *const slots = reinterpret_cast<unsigned *>(xh2 + 2 * G::kBufBytes);

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int h = lane >> 5;
    const int j = lane & 31;
    const ConvTile tile = conv_tile_of_block((long)a.c_out * a.c_in * (K * 4));     // two fp16 parts: 4 bytes per weight
    const int b = tile.b;
    const int mb = tile.mb;
    const int t0 = tile.nb * G::kBN;
    const int nchunks = (a.c_in + kConvCK - 1) / kConvCK;

    constexpr int kOob = kConvOob;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)b * a.x_bstride, ((long)(a.c_in - 1) * a.x_rstride + a.t_in) * 4);

    constexpr int kFull = G::kFull, kRem = G::kRem, kSingles = G::kSingles, kSets = G::kSets;
    constexpr int kInvalid = (int)0x80000000u;      // byte offset of a lane outside the tensor: + any row / chunk offset (< 2^31, conv_h2_offsets_ok) stays out of range
    int soff[kFull > 0 ? kFull : 1], soff1[kSingles > 0 ? kSingles : 1], lds1[kSingles > 0 ? kSingles : 1];
    float smask[kFull > 0 ? kFull : 1], smask1[kSingles > 0 ? kSingles : 1];
    {
        const __amdgpu_buffer_rsrc_t rm = make_rsrc(a.in_mask ? a.in_mask + (long)b * a.t_in : nullptr, a.in_mask ? (long)a.t_in * 4 : 0);
        const bool has_m = a.in_mask != nullptr;
#pragma unroll
        for (int i = 0; i < kFull; ++i) {
            const int e = tid + i * G::kThreads;
            const int half = e / G::kXW;
            const int col = e - half * G::kXW;
            const int gt = t0 - a.pad_left + col;
            const bool ok = (gt >= 0) && (gt < a.t_in);
            soff[i] = ok ? (int)(((long)(half * 8) * a.x_rstride + gt) * 4) : kInvalid;
            smask[i] = has_m ? ld_buf(rm, ok ? gt * 4 : kOob, 0) : 1.f;
        }
#pragma unroll
        for (int q = 0; q < kSingles; ++q) {
            // element s of the left-over items: channel s / kRem of item kFull * kThreads + s % kRem (consecutive lanes: consecutive columns)
            const int sidx = tid + q * G::kThreads;
            const bool valid = sidx < kRem * 8;
            const int ch = sidx / (kRem > 0 ? kRem : 1);
            const int e = kFull * G::kThreads + (sidx - ch * kRem);
            const int half = e / G::kXW;
            const int col = e - half * G::kXW;
            const int gt = t0 - a.pad_left + col;
            const bool ok = valid && (gt >= 0) && (gt < a.t_in);
            soff1[q] = ok ? (int)(((long)(half * 8 + ch) * a.x_rstride + gt) * 4) : kInvalid;
            smask1[q] = has_m ? ld_buf(rm, ok ? gt * 4 : kOob, 0) : 1.f;
            lds1[q] = valid ? half * (G::kXWp * 16) + col * 16 + ch * 2 : G::kXW * 16 + (tid & 7) * 2;    // idle lanes: the dump column
        }
    }
    const int row_bytes = (int)a.x_rstride * 4;
    float st8[kSets][kFull > 0 ? kFull : 1][8], st1[kSets][kSingles > 0 ? kSingles : 1];
    // staging step i of a chunk: i < kFull one 8-channel item, i == kFull the single elements.  Chunks beyond c_in read as zeros
    // (buffer range check), no memory traffic
    auto stage_load_step = [&](auto set, int i, int chunk) {
        const int cb = chunk * kConvCK * row_bytes;
        if (i < kFull) {
#pragma unroll
            // the chunk / row offset rides in the load's SCALAR offset (no VALU instruction per load): the range check of a raw buffer
            // access on gfx950 covers voffset + soffset (measured: scripts/ubench/soffset_range.hip), so chunks beyond c_in read zeros,
            // and an invalid lane's 2^31 + offset stays out of range and below 2^32 (conv_h2_offsets_ok)
            for (int c = 0; c < 8; ++c) st8[set][i][c] = ld_buf(rx, soff[i], cb + c * row_bytes);
        } else {
#pragma unroll
            for (int q = 0; q < kSingles; ++q) st1[set][q] = ld_buf(rx, soff1[q], cb);
        }
    };
    // mask + activation in place, and this wave's largest magnitude of the chunk -> its slot
    const bool has_in_mask = a.in_mask != nullptr;
    auto stage_act_max = [&](auto set, int parity) {
        float m = 0.f;
        if (has_in_mask) {                          // block-uniform: an unmasked launch pays no multiply per value
#pragma unroll
            for (int i = 0; i < kFull; ++i)
#pragma unroll
                for (int c = 0; c < 8; ++c) st8[set][i][c] *= smask[i];
#pragma unroll
            for (int q = 0; q < kSingles; ++q) st1[set][q] *= smask1[q];
        }
#pragma unroll
        for (int i = 0; i < kFull; ++i)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                st8[set][i][c] = conv_in_act(st8[set][i][c], a.in_act, a.in_slope);
                m = __builtin_fmaxf(m, __builtin_fabsf(st8[set][i][c]));
            }
#pragma unroll
        for (int q = 0; q < kSingles; ++q) {
            st1[set][q] = conv_in_act(st1[set][q], a.in_act, a.in_slope);
            m = __builtin_fmaxf(m, __builtin_fabsf(st1[set][q]));
        }
        const unsigned wmax = wave_max_u32(__builtin_bit_cast(unsigned, m));
        slots[parity * 8 + wave] = wmax;
    };
    auto chunk_max = [&](int parity) -> unsigned {       // after a barrier: the block's largest magnitude of that chunk
        const u32x4 s0 = *reinterpret_cast<const u32x4 *>(slots + parity * 8);
        unsigned m = max(max(s0.x, s0.y), max(s0.z, s0.w));
        if constexpr (WM * WN > 4) {
            const u32x4 s1 = *reinterpret_cast<const u32x4 *>(slots + parity * 8 + 4);
            m = max(m, max(max(s1.x, s1.y), max(s1.z, s1.w)));
        }
        return (unsigned)__builtin_amdgcn_readfirstlane((int)m);
    };
    auto stage_store_step = [&](auto set, int i, unsigned char *buf, float scale) {
        if (i < kFull) {
            const int e = tid + i * G::kThreads;
            const int half = e / G::kXW;
            const int col = e - half * G::kXW;
            unsigned pw[2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) conv_split2x2(st8[set][i][2 * c] * scale, st8[set][i][2 * c + 1] * scale, pw[0][c], pw[1][c]);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                u32x4 w;
                w.x = pw[q][0];
                w.y = pw[q][1];
                w.z = pw[q][2];
                w.w = pw[q][3];
                *reinterpret_cast<u32x4 *>(buf + q * G::kPartBytes + half * (G::kXWp * 16) + col * 16) = w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < kSingles; ++q) {
                const float x = st1[set][q] * scale;
                const _Float16 hi = (_Float16)x;
                const _Float16 lo = (_Float16)((x - (float)hi) * 2048.f);           // the arithmetic of conv_split2x2
                *reinterpret_cast<_Float16 *>(buf + lds1[q]) = hi;
                *reinterpret_cast<_Float16 *>(buf + G::kPartBytes + lds1[q]) = lo;
            }
        }
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, kSets - 1>;
    // K >= 7: item i is converted (and the same registers re-requested for the chunk after) at tap i; shorter kernels do all of
    // it at the top of the iteration (too few taps to spread over)
    constexpr bool kPipe = K >= 7;
    static_assert(!kPipe || G::kNStage < K - 1, "staging rounds must fit the taps");

    f32x16 accm[MI][NI], accx[MI][NI];
    const u32x4 *wp[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const long mtile = ((long)mb * WM + wm) * MI + mi;
        wp[mi] = reinterpret_cast<const u32x4 *>(a.w_h2) + mtile * ((long)nchunks * K * 2 * 64) + lane;
    }
    u32x4 a_cur[MI][2], a_nxt[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int q = 0; q < 2; ++q) a_cur[mi][q] = wp[mi][q * 64];
    const unsigned char *const table = reinterpret_cast<const unsigned char *>(a.w_h2) + conv_h2_table_offset(a.c_out, a.c_in, K);
    const int max_row_exp = reinterpret_cast<const H2RowTable *>(table)->max_row_exp;
    const float *const row_tab = reinterpret_cast<const float *>(table + sizeof(H2RowTable));     // [row][scale, unscale]

    // chunk n lives in register set n % kSets
#pragma unroll
    for (int i = 0; i < G::kNStage; ++i) stage_load_step(Set0{}, i, 0);
    if constexpr (kSets == 2) {
#pragma unroll
        for (int i = 0; i < G::kNStage; ++i) stage_load_step(Set1{}, i, 1);
    }
    bool folded = conv_acc_init<MODE, MI, NI, WM, WN>(accm, a, b, mb, t0, wm, wn, h, j);    // raw residual (or zeros)
    // (accx is an output of the main loop: its first product takes a zero C operand — an inline constant — instead of 16 NI MI v_mov)
    stage_act_max(Set0{}, 0);
    __syncthreads();
    int e_run = h2_exp_for(chunk_max(0));
#pragma unroll
    for (int i = 0; i < G::kNStage; ++i) stage_store_step(Set0{}, i, xh2, pow2f(e_run));
#pragma unroll
    for (int i = 0; i < G::kNStage; ++i) stage_load_step(Set0{}, i, kSets);       // chunk 1 (one set) / chunk 2 (two sets)
    if (folded) {
        // the residual enters the accumulators in their units: 2^(activation exponent + row exponent); kept out (and added by
        // the epilogue instead) in the corner where that factor could overflow
        if (e_run + max_row_exp <= kH2FoldMaxExp) {
            const float s = pow2f(e_run);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row0 = ((mb * WM + wm) * MI + mi) * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float rs = row_tab[2 * (row0 + (r & 3) + 8 * (r >> 2))];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) accm[mi][ni][r] = (accm[mi][ni][r] * s) * rs;
                }
            }
        } else {
            folded = false;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accm[mi][ni][r] = 0.f;
        }
    }
    stage_act_max(Set1{}, 1);             // chunk 1
    __syncthreads();

    const int bbyte = h * (G::kXWp * 16) + (wn * (32 * NI) + j) * 16;
    // iteration c: chunk c + 1 (registers `s1`, maximum known since the last barrier) is converted into the other LDS buffer and its
    // registers re-requested for chunk c + 1 + kSets; chunk c's MFMAs; chunk c + 2 (registers `s2`) gets mask / activation / maximum
    constexpr f32x16 kZero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto iteration = [&](auto first, int c, auto s1, auto s2) {
        const unsigned char *cur = xh2 + (c & 1) * G::kBufBytes + bbyte;
        unsigned char *const nxt = xh2 + ((c + 1) & 1) * G::kBufBytes;
        // exponent of chunk c + 1 (its maximum is in the slots since the last barrier)
        int e_next = e_run;
        if (c + 1 < nchunks) {
            const unsigned m = chunk_max((c + 1) & 1);
            if ((int)(m >> 23) - 127 + e_run >= kH2LimitExp) e_next = h2_exp_for(m);
        }
        const float s_next = pow2f(e_next);
        if constexpr (!kPipe) {
#pragma unroll
            for (int i = 0; i < G::kNStage; ++i) stage_store_step(s1, i, nxt, s_next);
#pragma unroll
            for (int i = 0; i < G::kNStage; ++i) stage_load_step(s1, i, c + 1 + kSets);
        }
        // NI = 1 (the eight-wave mid-size tile): a tap is two LDS fragment reads and three MFMAs that wait for them — 96 cycles of matrix
        // work behind a ~200-cycle read, every tap, and with one block per CU no third wave to cover it.  The activation fragments
        // are requested a tap ahead too, like the weight fragments (TTSAMD_H2_BPREFETCH=0 builds the read-then-use form).
#ifndef TTSAMD_H2_BPREFETCH
#define TTSAMD_H2_BPREFETCH 1
#endif
        constexpr bool kBP = (NI == 1) && TTSAMD_H2_BPREFETCH;
        u32x4 bq_n[2];
        if constexpr (kBP) {
#pragma unroll
            for (int q = 0; q < 2; ++q) bq_n[q] = *reinterpret_cast<const u32x4 *>(cur + q * G::kPartBytes);
        }
#pragma unroll
        for (int tap = 0; tap < K; ++tap) {
            const long g = ((tap + 1 < K) ? ((long)c * K + tap + 1) : ((long)(c + 1) * K)) * (2 * 64);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int q = 0; q < 2; ++q) a_nxt[mi][q] = wp[mi][g + q * 64];
            u32x4 bq_c[2];
            if constexpr (kBP) {
#pragma unroll
                for (int q = 0; q < 2; ++q) bq_c[q] = bq_n[q];
                if (tap + 1 < K) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) bq_n[q] = *reinterpret_cast<const u32x4 *>(cur + q * G::kPartBytes + ((tap + 1) * D) * 16);
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch a whole tap ahead of its use
            if constexpr (kPipe) {
                if (tap < G::kNStage) {
                    stage_store_step(s1, tap, nxt, s_next);
                    stage_load_step(s1, tap, c + 1 + kSets);
                }
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                u32x4 bq[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if constexpr (kBP) bq[q] = bq_c[q];
                    else bq[q] = *reinterpret_cast<const u32x4 *>(cur + q * G::kPartBytes + (ni * 32 + tap * D) * 16);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const bool zero = decltype(first)::value && tap == 0;
                    accx[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_cur[mi][1]), __builtin_bit_cast(f16x8, bq[0]), zero ? kZero : accx[mi][ni], 0, 0, 0);
                    accx[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_cur[mi][0]), __builtin_bit_cast(f16x8, bq[1]), accx[mi][ni], 0, 0, 0);
                    accm[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_cur[mi][0]), __builtin_bit_cast(f16x8, bq[0]), accm[mi][ni], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int q = 0; q < 2; ++q) a_cur[mi][q] = a_nxt[mi][q];
        }
        stage_act_max(s2, c & 1);         // chunk c + 2 (same parity as c)
        if (e_next != e_run) {            // block-uniform, rare: the accumulators follow the running exponent (exact power of two)
            const float f = pow2f(e_next - e_run < -126 ? -126 : e_next - e_run);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        accm[mi][ni][r] *= f;
                        accx[mi][ni][r] *= f;
                    }
            e_run = e_next;
        }
        __syncthreads();
    };
    iteration(std::true_type{}, 0, Set1{}, Set0{});           // (one set: Set1 == Set0)
    if constexpr (kSets == 2) {
        for (int c = 1; c < nchunks; c += 2) {
            iteration(std::false_type{}, c, Set0{}, Set1{});
            if (c + 1 >= nchunks) break;
            iteration(std::false_type{}, c + 1, Set1{}, Set0{});
        }
    } else {
        for (int c = 1; c < nchunks; ++c) iteration(std::false_type{}, c, Set0{}, Set0{});
    }

    // the two accumulators meet and leave the scaled units (activation exponent, then the row's), then the shared epilogue
    {
        const float us = pow2f(-e_run);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row0 = ((mb * WM + wm) * MI + mi) * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ru = row_tab[2 * (row0 + (r & 3) + 8 * (r >> 2)) + 1];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) accm[mi][ni][r] = (__builtin_fmaf(accx[mi][ni][r], 1.f / 2048.f, accm[mi][ni][r]) * us) * ru;   // (x 2^-11 exact: same bits)
            }
        }
    }
    conv_epilogue<MODE, MI, NI, WM, WN>(accm, b, mb, t0, wm, wn, h, j, folded);
}

template <int K, int D, int MI, int NI, int WM, int WN, int MODE>
int conv1d_h2_launch_cfg(const ttsamd_conv1d_args &a, hipStream_t st)
{
    using G = ConvGeomH2<K, D, MI, NI, WM, WN>;
    auto kern = conv1d_h2_kernel<K, D, MI, NI, WM, WN, MODE>;
    static std::atomic<unsigned long long> lds_attr_done{0};
    TTSAMD_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), (int)G::kLdsBytes, lds_attr_done));
    const int mtiles = (a.c_out + 31) / 32;
    const int mblocks = (mtiles + MI * WM - 1) / (MI * WM);
    const int nblocks = (a.t_out + G::kBN - 1) / G::kBN;
    hipLaunchKernelGGL(kern, dim3(nblocks, mblocks, a.batch), dim3(G::kThreads), G::kLdsBytes, st, a);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

// the large-grid tiles of conv1d_x3_launch_tiles, three-product arithmetic
template <int K, int D, int MODE>
int conv1d_h2_launch_tiles(const ttsamd_conv1d_args &a, hipStream_t st)
{
    const int mtiles = (a.c_out + 31) / 32;
    // Awkward lengths (the headline's T = 257 text columns and 770 frames are 2 x 128 + 1 and 6 x 128 + 2): a tile half as wide when
    // that removes >= 10 % of the padded columns the launch computes.  TTSAMD_H2_ADAPT_TILES=0: the fixed tiles (A/B switch).
    static const bool adapt = !(getenv("TTSAMD_H2_ADAPT_TILES") && getenv("TTSAMD_H2_ADAPT_TILES")[0] == '0');
    auto padded = [&](int bn) { return (long)((a.t_out + bn - 1) / bn) * bn; };
    // (A tile half as wide for launches of less than two rounds of the wide one — the flow convs' 672 blocks — was measured and is
    // not kept: gate conv 75 -> 71.5 us, res/skip 1x1 46 -> 50, whole step 48.81 -> 48.95 ms; profiles/r06_xlocal_ab.txt.)
    if (mtiles % 4 == 0) {
        if (adapt && padded(64) * 10 <= padded(128) * 9) return conv1d_h2_launch_cfg<K, D, 1, 2, 4, 1, MODE>(a, st);      // 128 rows x 64 columns
        return conv1d_h2_launch_cfg<K, D, TTSAMD_X3_CFG128, MODE>(a, st);
    }
    if (mtiles % 2 == 0) {
        if (adapt && padded(128) * 10 <= padded(256) * 9) return conv1d_h2_launch_cfg<K, D, 1, 2, 2, 2, MODE>(a, st);     // 64 rows x 128 columns
        return conv1d_h2_launch_cfg<K, D, TTSAMD_X3_CFG64, MODE>(a, st);
    }
    return conv1d_h2_launch_cfg<K, D, 1, 2, 1, 4, MODE>(a, st);
}

// Mid-size grids (48 .. 128 of the 128 x 128-class blocks: a lone request's 256-channel stage, 98 of them -> 200 blocks of 128 x 64 on
// 256 CUs).  A CU holds at most one such block, so its four waves walk the whole K loop alone on their SIMDs; the same tile on EIGHT
// waves (32 x 32 each) puts two waves on a SIMD and halves each one's MFMA / staging chain: VITS B = 1 request 3.31 -> 3.25 ms same
// box (profiles/r06_w8_mid_ab.txt).  The 128 x 128 tile on eight waves for launches below one round (the 128-channel stage of that
// request: 385 blocks) was measured with it and LOSES (3.31 -> 3.35 ms).  TTSAMD_H2_MID_W8=0: four waves (A/B switch).
template <int K, int D, int MODE>
int conv1d_h2_launch_mid(const ttsamd_conv1d_args &a, hipStream_t st)
{
    static const bool w8 = !(getenv("TTSAMD_H2_MID_W8") && getenv("TTSAMD_H2_MID_W8")[0] == '0');
    if (w8) return conv1d_h2_launch_cfg<K, D, 1, 1, 4, 2, MODE>(a, st);
    return conv1d_h2_launch_cfg<K, D, 1, 2, 4, 1, MODE>(a, st);
}

}  // namespace ttsamd
