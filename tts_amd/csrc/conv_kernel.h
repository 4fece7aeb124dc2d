// Conv1d as implicit GEMM on the fp32-input MFMA of gfx950 (v_mfma_f32_32x32x2_f32).
//
//   y[b,co,t] = epilogue( bias[co] + sum_{ci,k} W[co,ci,k] * act(x[b,ci,t + k*D - pad_left]) )
//
// GEMM view: M = c_out rows, N = time, reduction = (ci, tap).  One MFMA consumes two reduction
// indices: lanes 0-31 carry channel 2p, lanes 32-63 channel 2p+1, same tap (kernel sizes are odd, so
// channels are paired, not taps).
//   * A (weights): pre-packed at load time in exact fragment order [m-tile][k-step group][lane][4],
//     so every wave fetches its fragments with one fully coalesced 16-byte-per-lane load per four
//     k-steps straight from L2 (a c_out-tile's weights are re-used by every time tile; measured
//     need ~2-4 B/clk/CU, far below L2 bandwidth) — no LDS round trip, software-prefetched one
//     group ahead.
//   * B (activations): a [16 channels][BN + halo] tile is staged once per channel chunk through
//     registers into LDS (double buffered, one barrier per chunk); leaky-ReLU / input masking are
//     applied ONCE per staged element, not per tap.  B fragments are `ds_read_b32` with a single
//     base VGPR + compile-time immediates (K, D are template parameters): lanes read 32 consecutive
//     floats per half-wave => bank-conflict free.
//   * D: 32x32 fp32 accumulators stay in registers for the whole reduction; the epilogue fuses
//     bias, activation, residual add, MRF accumulation, masking, the WaveNet tanh*sigmoid gate,
//     the mean-only affine coupling and the polyphase "pixel shuffle" of ConvTranspose1d.
// fp32 MFMA is bit-for-bit an fmaf chain (one rounding per product), i.e. the same arithmetic
// class as the reference's fp32 CPU conv; only the summation order differs.
#pragma once
#include "common.h"
#include "pack_layout.h"

namespace ttsamd {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4s = __attribute__((ext_vector_type(4))) float;
using f32x2u = __attribute__((ext_vector_type(2), aligned(4))) float;   // 8-byte vector at 4-byte alignment

// kConvCK = 16 input channels per LDS chunk (8 channel pairs): pack_layout.h
constexpr int kConvOob = kBufOob;

struct ConvTileCfg {
    int mi, ni, wm, wn;
};

template <int K, int D, int MI, int NI, int WM, int WN>
struct ConvGeom {
    static constexpr int kThreads = 64 * WM * WN;
    static constexpr int kBM = 32 * MI * WM;
    static constexpr int kBN = 32 * NI * WN;
    static constexpr int kHalo = (K - 1) * D;
    static constexpr int kXW = kBN + kHalo;                 // staged columns per channel
    static constexpr int kStageElems = kConvCK * kXW;
    static constexpr int kNStage = (kStageElems + kThreads - 1) / kThreads;
    static constexpr int kGroupsPerChunk = (kConvCK / 2) * K / 4;  // k-step groups (of 4) per chunk
    static constexpr size_t kLdsBytes = (size_t)2 * kStageElems * sizeof(float);
};

// Block -> tile assignment.  The hardware deals workgroups to the 8 XCDs round-robin by linear id (x fastest), each XCD
// with its own 4 MiB L2.  With the plain (x = time tile, y = m-block, z = item) order every L2 sees every m-block's
// weight stream (4.3 MB for a 256x256 k=11 layer: it thrashes, PMC: 1.7 GB fetched per launch against 0.5 GB of
// activations) and neighbouring time tiles — which share their halo columns — sit in different L2s.  Remapped: XCD c
// works on m-block c % mblocks only and walks a contiguous range of (item, time tile) pairs.
struct ConvTile {
    int nb, mb, b;   // time tile, m-block, batch item
};
// Round 6, `wbytes` (the launch's packed weight image, bytes; < 0: not given): a SMALL image (polyphase ups[1] 2.1 MB, the flow /
// text convs 0.3-1.5 MB) fits every L2 next to the activations, and then it is the x tile that should be fetched once: all
// m-blocks of an (item, time tile) pair run back to back on ONE XCD ("x-local"; the weights-local order made every XCD fetch the
// whole of x: 8 x at ups[1]).  Larger images keep the weights-local order, now also for 16, 24 ... m-blocks (ups[0]: two per XCD).
#ifndef TTSAMD_XCD_WIDE
#define TTSAMD_XCD_WIDE 1      // 0 (with TTSAMD_XLOCAL_WBYTES=-1): the block order of rounds 2-5, for A/B builds
#endif
#ifndef TTSAMD_XLOCAL_WBYTES
#define TTSAMD_XLOCAL_WBYTES (2560 * 1024)
#endif
__device__ __forceinline__ ConvTile conv_tile_of_block(long wbytes = -1)
{
    ConvTile t{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const unsigned total = gx * gy * gz;
    if ((total & 7u) != 0) return t;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = lin & 7u, i = lin >> 3, cnt = total >> 3;
    const unsigned pairs = gx * gz;
    if (gy > 1 && wbytes >= 0 && wbytes <= (long)TTSAMD_XLOCAL_WBYTES && (pairs & 7u) == 0) {
        // x-local: XCD c walks a contiguous eighth of the (item, time tile) pairs, every m-block of a pair in a row
        const unsigned q = i / gy;
        t.mb = (int)(i - q * gy);
        const unsigned u = xcd * (pairs >> 3) + q;
        t.b = (int)(u / gx);
        t.nb = (int)(u - (unsigned)t.b * gx);
    } else if (gy == 1 || gy == 2 || gy == 4 || gy == 8) {
        t.mb = (int)(xcd % gy);
        const unsigned u = (xcd / gy) * cnt + i;     // index among the (item, time tile) pairs of this m-block
        t.b = (int)(u / gx);
        t.nb = (int)(u - (unsigned)t.b * gx);
    } else if (TTSAMD_XCD_WIDE && (gy & 7u) == 0) {
        const unsigned per = gy >> 3;                // m-blocks per XCD
        const unsigned u = i / per;
        t.mb = (int)(xcd * per + (i - u * per));
        t.b = (int)(u / gx);
        t.nb = (int)(u - (unsigned)t.b * gx);
    }
    return t;
}

// leaky ReLU as max(v, v * slope): for 0 <= slope <= 1 (checked on the host: ttsamd_conv1d / ttsamd_resblock_pair refuse
// other slopes) this is v > 0 ? v : v * slope value for value, signed zeros included — one v_max instead of a compare + select
// (the instruction itself: through fmaxf() hipcc first canonicalises both operands — a v_max v, v, v each — for sNaN inputs)
__device__ __forceinline__ float conv_lrelu(float v, float slope)
{
    float r;
    const float vs = v * slope;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(vs));
    return r;
}
// No activation = slope 1 (v * 1 and max(v, v) are exact): the choice is one scalar select per launch, not a compare + select
// per staged element.
__device__ __forceinline__ float conv_in_act(float v, int act, float slope)
{
    return conv_lrelu(v, (act == TTSAMD_ACT_LRELU) ? slope : 1.f);
}

// Accumulator initial value, shared by the fp32-MFMA and the split-bf16 kernels.
template <int MODE, int MI, int NI, int WM, int WN>
__device__ __forceinline__ bool conv_acc_init(f32x16 (&acc)[MI][NI], const ttsamd_conv1d_args &a, int b, int mb, int t0, int wm,
                                              int wn, int h, int j)
{
    constexpr int kOob = kConvOob;
    // NORMAL mode without an output activation: the residual operand is folded into the
    // accumulators' INITIAL value (D = A.B + C with C = res).  Their HBM latency then overlaps the
    // prologue's weight / activation fetches instead of being exposed after the main loop, at zero register cost
    // (measured: the post-loop residual read cost +20..75 % per launch on the C<=64 stages).
    bool folded = false;
    if constexpr (MODE == TTSAMD_CONV_NORMAL) folded = (a.out_act == TTSAMD_ACT_NONE) && a.res;
    if (folded) {
        const __amdgpu_buffer_rsrc_t rr = make_rsrc(a.res + (long)b * a.res_bstride,
                                                    ((long)(a.c_out - 1) * a.res_rstride + a.t_out) * 4);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row0 = ((mb * WM + wm) * MI + mi) * 32;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int t = t0 + wn * (32 * NI) + ni * 32 + j;
                const int vo = (t < a.t_out) ? (int)(((long)(4 * h) * a.res_rstride + t) * 4) : kOob;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rb = row0 + (r & 3) + 8 * (r >> 2);                 // wave-uniform part of the row
                    const int vr = (rb + 4 * h < a.c_out) ? vo : kOob;
                    acc[mi][ni][r] = ld_buf(rr, vr, rb * (int)a.res_rstride * 4);
                }
            }
        }
    } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    }

    return folded;
}

// Fused epilogue (bias, activation, residual, MRF accumulate, masks, gate, couplings, polyphase shuffle), shared by the
// fp32-MFMA and the split-bf16 kernels: the 32x32 accumulator layout of every gfx950 MFMA is the same.
// Wave-uniform part of the accumulator row a lane holds in register r of a 32x32 tile (the lane adds 4 * h): the whole tile
// (NR = 16), or — the small-grid one-shot kernel spreads a tile's epilogue over four waves — quarter `rq` of it (NR = 4):
// registers {2q, 2q+1, 2q+8, 2q+9}, i.e. rows 2(q&1) + 8(q>>1) + {0, 1, 16, 17}: register r and r + NR/2 are 16 rows apart
// in both forms, which is what the paired-row modes need.
template <int NR>
__device__ __forceinline__ int conv_erow(int r, int rq)
{
    if constexpr (NR == 16) return (r & 3) + 8 * (r >> 2);
    return 2 * (rq & 1) + 8 * (rq >> 1) + (r & 1) + 16 * (r >> 1);
}

template <int MODE, int MI, int NI, int WM, int WN, int NR = 16>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[MI][NI], int b, int mb, int t0, int wm, int wn, int h, int j, bool folded,
                                              int rq = 0)
{
    static_assert(NR == 16 || NR == 4, "whole tile or a quarter");
    static_assert(NR == 16 || MODE != TTSAMD_CONV_SHUFFLE, "the polyphase stores take whole tiles");
    constexpr int kOob = kConvOob;
    // ---- epilogue --------------------------------------------------------------------------
    // MODE is a template parameter (each fusion gets its own lean kernel), and the epilogue-only
    // arguments are read from the kernarg segment HERE, behind an opaque barrier, so that they do
    // not sit in SGPRs (or spill) across the MFMA main loop.
    const ttsamd_conv1d_args __attribute__((address_space(4))) *ep =
        (const ttsamd_conv1d_args __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ep) : : "memory");
    // Every argument the epilogue can need is requested HERE, as one batch of scalar loads behind one wait (the empty asm
    // below pins them to this point).  Left to itself the compiler sinks each load next to its first use, behind the
    // wave-uniform branch that guards it: ten dependent s_load -> s_waitcnt -> branch hops in a row, ~2 us of a small
    // launch's ~8 us block time (scripts/phase_clocks.py).
    float *const y = ep->y;
    const long y_bs = ep->y_bstride, y_rs = ep->y_rstride;
    const int c_out = ep->c_out, t_out = ep->t_out;
    const float *const bias = ep->bias;
    const float *const rbias0 = ep->row_bias;
    const float *const omask0 = ep->out_mask;
    const float *const res0 = ep->res;
    const long res_bs = ep->res_bstride, res_rs = ep->res_rstride;
    const float *const accum0 = ep->accum;
    const long accum_bs = ep->accum_bstride, accum_rs = ep->accum_rstride;
    const int out_act = ep->out_act;
    const float out_div = ep->out_div;
    constexpr bool kUsesSplit = (MODE == TTSAMD_CONV_RES_SKIP || MODE == TTSAMD_CONV_COUPLE_AFFINE || MODE == TTSAMD_CONV_COUPLE_AFFINE_FWD || MODE == TTSAMD_CONV_COUPLE_AFFINE_MIX);
    const int split_row = kUsesSplit ? ep->split_row : 0;
    float *const y2_0 = (MODE == TTSAMD_CONV_RES_SKIP || MODE == TTSAMD_CONV_COUPLE_AFFINE_MIX) ? ep->y2 : nullptr;
    const long y2_bs = (MODE == TTSAMD_CONV_RES_SKIP) ? ep->y2_bstride : 0, y2_rs = (MODE == TTSAMD_CONV_RES_SKIP) ? ep->y2_rstride : 0;
    const int shuffle_u = (MODE == TTSAMD_CONV_SHUFFLE) ? ep->shuffle_u : 1;
    const int shuffle_pad = (MODE == TTSAMD_CONV_SHUFFLE) ? ep->shuffle_pad : 0;
    const int shuffle_t_out = (MODE == TTSAMD_CONV_SHUFFLE) ? ep->shuffle_t_out : 0;
    asm volatile("" : : "s"(y), "s"(y_bs), "s"(y_rs), "s"(c_out), "s"(t_out), "s"(bias), "s"(rbias0), "s"(omask0), "s"(res0),
                 "s"(res_bs), "s"(res_rs), "s"(accum0), "s"(accum_bs), "s"(accum_rs), "s"(out_act),
                 "s"(__builtin_bit_cast(int, out_div)), "s"(split_row), "s"(y2_0), "s"(y2_bs), "s"(y2_rs), "s"(shuffle_u),
                 "s"(shuffle_pad), "s"(shuffle_t_out));
    const float *rbias = rbias0 ? rbias0 + (long)b * c_out : nullptr;
    const float *omask = omask0 ? omask0 + (long)b * t_out : nullptr;
    const float *res = res0 ? res0 + (long)b * res_bs : nullptr;

    if constexpr (MODE == TTSAMD_CONV_GATE || MODE == TTSAMD_CONV_COUPLE_AFFINE || MODE == TTSAMD_CONV_COUPLE_AFFINE_FWD ||
                  MODE == TTSAMD_CONV_COUPLE_AFFINE_MIX) {
        // Paired rows live in ONE 32-row tile (round 4; before: in two neighbouring tiles, which tied these modes to MI = 2):
        // packed rows [32m, 32m+16) = the first halves (tanh / t) of output channels [16m, 16m+16), rows [32m+16, 32m+32) the
        // second halves (sigmoid / s) — in the accumulator layout register r (< NR/2) and register r + NR/2 of the same lane.
        // Branch-free passes like the unpaired modes below: the first version took a divergent branch per element
        // with its bias / row-bias / coupling-operand loads INSIDE it — sixteen dependent global round trips per lane,
        // 12 200 of a 29 700-cycle block at the single-sentence shapes (scripts/phase_clocks.py, 192 -> 384, k = 5, T = 159).
        constexpr bool gate = (MODE == TTSAMD_CONV_GATE);
        constexpr int NP = NR / 2;                         // pairs per lane and tile
        const int nvalid = gate ? c_out / 2 : split_row;  // output channels
        const int y_rs4 = (int)y_rs * 4, res_rs4 = (int)res_rs * 4;
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(y + (long)b * y_bs, ((long)(nvalid - 1) * y_rs + t_out) * 4);
        const __amdgpu_buffer_rsrc_t rres = make_rsrc(res, (!gate && res) ? ((long)(nvalid - 1) * res_rs + t_out) * 4 : 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int mt = (mb * WM + wm) * MI + mi;
            float b0[NP], b1[NP];     // bias (+ per-item row bias) of this lane's row pairs: packed rows 32 mt + i and + 16
            bool rok[NP];
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                const int i = conv_erow<NR>(r, rq) + 4 * h;
                rok[r] = (mt * 32 + i + 16 < c_out) && (mt * 16 + i < nvalid);
                b0[r] = 0.f;
                b1[r] = 0.f;
            }
            if (bias) {
#pragma unroll
                for (int r = 0; r < NP; ++r) {
                    const int prow = mt * 32 + conv_erow<NR>(r, rq) + 4 * h;
                    b0[r] = bias[rok[r] ? prow : 0];
                    b1[r] = bias[rok[r] ? prow + 16 : 0];
                }
            }
            if (rbias) {
                float c0[NP], c1[NP];
#pragma unroll
                for (int r = 0; r < NP; ++r) {
                    const int prow = mt * 32 + conv_erow<NR>(r, rq) + 4 * h;
                    c0[r] = rbias[rok[r] ? prow : 0];
                    c1[r] = rbias[rok[r] ? prow + 16 : 0];
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < NP; ++r) {   // (v + bias) + row_bias
                        acc[mi][ni][r] = (acc[mi][ni][r] + b0[r]) + c0[r];
                        acc[mi][ni][r + NP] = (acc[mi][ni][r + NP] + b1[r]) + c1[r];
                    }
            } else {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < NP; ++r) {
                        acc[mi][ni][r] += b0[r];
                        acc[mi][ni][r + NP] += b1[r];
                    }
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int t = t0 + wn * (32 * NI) + ni * 32 + j;
                const bool tv = t < t_out;
                float e1[NP];
#pragma unroll
                for (int r = 0; r < NP; ++r) e1[r] = 0.f;
                float om = 1.f;
                if constexpr (!gate) {
                    om = omask ? omask[tv ? t : 0] : 1.f;
#pragma unroll
                    for (int r = 0; r < NP; ++r) {       // the coupled half x1 (may be the very rows y overwrites: read first)
                        const int oc = mt * 16 + conv_erow<NR>(r, rq);             // wave-uniform part of the output channel
                        e1[r] = ld_buf(rres, (tv && rok[r]) ? (4 * h * res_rs4 + t * 4) : kOob, oc * res_rs4);
                    }
                }
                if constexpr (MODE == TTSAMD_CONV_COUPLE_AFFINE_MIX) {
                    // the coupled values stay in registers; InvConvNear^-1 mixes channel pairs (2i, 2i+1) of the untouched half
                    // (read here) with the same pair of the coupled half (registers r, r+1: consecutive rows of one lane), then
                    // ActNorm^-1 — the operations of glow_invconv_actnorm_kernel in its order
                    static_assert(NP % 2 == 0, "channel pairs");
                    const float *const mixp = y2_0;                      // [16] w_inv, [C] bias, [C] logs
                    const int Cc = 2 * nvalid;
                    float w[4][4];
#pragma unroll
                    for (int a4 = 0; a4 < 4; ++a4)
#pragma unroll
                        for (int b4 = 0; b4 < 4; ++b4) w[a4][b4] = mixp[a4 * 4 + b4];
                    const float *const x0base = res - (long)nvalid * res_rs;       // the untouched half sits nvalid rows before
                    float *const y0base = y + (long)b * y_bs - (long)nvalid * y_rs;
#pragma unroll
                    for (int r = 0; r < NP; r += 2) {
                        const int oc = mt * 16 + conv_erow<NR>(r, rq) + 4 * h;     // even channel of the pair
                        const bool ok = tv && rok[r] && rok[r + 1];
                        float v[4];
                        v[0] = ok ? x0base[(long)oc * res_rs + t] : 0.f;
                        v[1] = ok ? x0base[(long)(oc + 1) * res_rs + t] : 0.f;
                        v[2] = (e1[r] - acc[mi][ni][r]) * expf(-acc[mi][ni][r + NP]) * om;
                        v[3] = (e1[r + 1] - acc[mi][ni][r + 1]) * expf(-acc[mi][ni][r + 1 + NP]) * om;
#pragma unroll
                        for (int go = 0; go < 4; ++go) {
                            float z = 0.f;
#pragma unroll
                            for (int g = 0; g < 4; ++g) z += w[go][g] * v[g];
                            const int ch = (go >> 1) * nvalid + oc + (go & 1);
                            z *= om;
                            z = (z - mixp[16 + (ok ? ch : 0)]) * expf(-mixp[16 + Cc + (ok ? ch : 0)]) * om;
                            if (ok) {
                                if (go < 2) y0base[(long)(oc + go) * y_rs + t] = z;
                                else y[(long)b * y_bs + (long)(oc + go - 2) * y_rs + t] = z;
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < NP; ++r) {
                        const float v0 = acc[mi][ni][r], v1 = acc[mi][ni][r + NP];
                        float o;
                        if constexpr (gate) {
                            o = tanhf(v0) * (1.f / (1.f + expf(-v1)));
                        } else if constexpr (MODE == TTSAMD_CONV_COUPLE_AFFINE) {
                            o = (e1[r] - v0) * expf(-v1) * om;
                        } else {
                            o = (v0 + expf(v1) * e1[r]) * om;
                        }
                        const int oc = mt * 16 + conv_erow<NR>(r, rq);
                        st_buf(ry, o, (tv && rok[r]) ? (4 * h * y_rs4 + t * 4) : kOob, oc * y_rs4);
                    }
                }
            }
        }
    } else {
        // Branch-free passes: every optional operand is fetched by a whole pass of buffer loads behind ONE
        // wave-uniform branch (never a branch + wait per element); validity lives in the offsets (kOob).
        const int split = (MODE == TTSAMD_CONV_RES_SKIP) ? split_row : 0;
        const int y_rs4 = (int)y_rs * 4, res_rs4 = (int)res_rs * 4, acc_rs4 = (int)accum_rs * 4;
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(
            y + (long)b * y_bs, (MODE == TTSAMD_CONV_SHUFFLE)
                                    ? ((long)((c_out - 1) / shuffle_u) * y_rs + shuffle_t_out) * 4
                                    : ((long)((MODE == TTSAMD_CONV_RES_SKIP ? split : c_out) - 1) * y_rs + t_out) * 4);
        const __amdgpu_buffer_rsrc_t rres = make_rsrc(res, res ? ((long)(c_out - 1) * res_rs + t_out) * 4 : 0);
        const __amdgpu_buffer_rsrc_t racc = make_rsrc(
            accum0 ? accum0 + (long)b * accum_bs : nullptr,
            accum0 ? ((long)(c_out - split - 1) * accum_rs + t_out) * 4 : 0);
        const __amdgpu_buffer_rsrc_t ry2 = make_rsrc(
            (MODE == TTSAMD_CONV_RES_SKIP) ? y2_0 + (long)b * y2_bs : nullptr,
            (MODE == TTSAMD_CONV_RES_SKIP) ? ((long)(c_out - split - 1) * y2_rs + t_out) * 4 : 0);
        const int y2_rs4 = (MODE == TTSAMD_CONV_RES_SKIP) ? (int)y2_rs * 4 : 0;
        const bool has_accum = accum0 != nullptr;
        const bool all_rows = (c_out & 31) == 0;
        // polyphase ConvTranspose with a stride that is a multiple of 4 (HiFiGAN ups[0], ups[1]: 8): 16-byte stores.  Groups
        // start at sample indices that are multiples of 4, so with shuffle_t_out % 4 == 0 no group straddles the row end
        // (a C-ABI caller's odd output length takes the dword path below instead of losing its last partial group)
        const bool shuffle_vec = (MODE == TTSAMD_CONV_SHUFFLE) && (shuffle_u % 4 == 0) && (shuffle_pad % 4 == 0) &&
                                 (c_out % 4 == 0) && ((y_rs & 3) == 0) && ((y_bs & 3) == 0) && ((shuffle_t_out & 3) == 0) &&
                                 ((reinterpret_cast<unsigned long long>(y) & 15ull) == 0);
        const bool shuffle_vec2 = (MODE == TTSAMD_CONV_SHUFFLE) && shuffle_u == 2 && shuffle_pad == 1 && (c_out % 2 == 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row0 = ((mb * WM + wm) * MI + mi) * 32;
            const bool lower = (MODE == TTSAMD_CONV_RES_SKIP) && (row0 < split);   // res rows vs skip rows
            float radd[NR];   // bias (+ per-item row bias) of this lane's 16 rows of the m-tile
            // through buffer resources: rows >= c_out (and an absent operand: zero-length resource) read as 0 by the range check —
            // no 64-bit address arithmetic, no clamp per row (round 6: 64 VALU instructions per m-tile, on a chip where every
            // VALU instruction costs the matrix pipe an eighth of an MFMA's issue time)
            {
                const __amdgpu_buffer_rsrc_t rbi = make_rsrc(bias, bias ? (long)c_out * 4 : 0);
                const __amdgpu_buffer_rsrc_t rrb = make_rsrc(rbias, rbias ? (long)c_out * 4 : 0);
#pragma unroll
                for (int r = 0; r < NR; ++r) radd[r] = ld_buf(rbi, 16 * h, (row0 + conv_erow<NR>(r, rq)) * 4);
                if (rbias) {
#pragma unroll
                    for (int r = 0; r < NR; ++r) radd[r] += ld_buf(rrb, 16 * h, (row0 + conv_erow<NR>(r, rq)) * 4);
                }
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                // one 32x32 tile at a time: keeps the epilogue's live registers below the main loop's
                __builtin_amdgcn_sched_barrier(0);
                const int t = t0 + wn * (32 * NI) + ni * 32 + j;
                const bool tv = t < t_out;
                const float om = omask ? omask[tv ? t : 0] : 1.f;
                float e1[NR], e2[NR];   // optional operands of this 32x32 tile
#pragma unroll
                for (int r = 0; r < NR; ++r) { e1[r] = 0.f; e2[r] = 0.f; }
                const bool need_res = (MODE == TTSAMD_CONV_COUPLE) || (MODE == TTSAMD_CONV_RES_SKIP && lower) ||
                                      (MODE == TTSAMD_CONV_NORMAL && res && !folded);
                if (need_res) {
                    const int vo = tv ? (4 * h * res_rs4 + t * 4) : kOob;
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const int rb = row0 + conv_erow<NR>(r, rq);
                        e1[r] = ld_buf(rres, (rb + 4 * h < c_out) ? vo : kOob, rb * res_rs4);
                    }
                }
                const bool need_acc = has_accum && (MODE == TTSAMD_CONV_NORMAL || (MODE == TTSAMD_CONV_RES_SKIP && !lower));
                if (need_acc) {
                    const int vo = tv ? (4 * h * acc_rs4 + t * 4) : kOob;
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const int rb = row0 - split + conv_erow<NR>(r, rq);
                        e2[r] = ld_buf(racc, (rb + 4 * h + split < c_out) ? vo : kOob, rb * acc_rs4);
                    }
                }
                const int voy = tv ? (4 * h * y_rs4 + t * 4) : kOob;
                const int voy2 = tv ? (4 * h * y2_rs4 + t * 4) : kOob;
                const bool has_div = out_div != 0.f;
                if constexpr (MODE == TTSAMD_CONV_NORMAL) {
                    // bias + activation in place on the accumulators; out_act is wave-uniform: its branches sit outside the
                    // unrolled element loop
#pragma unroll
                    for (int r = 0; r < NR; ++r) acc[mi][ni][r] += radd[r];
                    if (out_act == TTSAMD_ACT_RELU) {
#pragma unroll
                        for (int r = 0; r < NR; ++r) acc[mi][ni][r] = fmaxf(acc[mi][ni][r], 0.f);
                    } else if (out_act == TTSAMD_ACT_TANH) {
#pragma unroll
                        for (int r = 0; r < NR; ++r) acc[mi][ni][r] = tanhf(acc[mi][ni][r]);
                    }
                    // residual, accumulate, mask, division: each behind ONE wave-uniform branch around its whole pass (inside
                    // the element loop hipcc evaluates the IEEE division sequence — 12 VALU instructions — for every element of
                    // every launch and selects afterwards).  Same operations in the same order as before.
                    if (need_res) {                                                    // (absent / folded: no pass of + 0)
#pragma unroll
                        for (int r = 0; r < NR; ++r) acc[mi][ni][r] += e1[r];
                    }
                    if (need_acc) {
#pragma unroll
                        for (int r = 0; r < NR; ++r) acc[mi][ni][r] = e2[r] + acc[mi][ni][r];
                    }
                    if (omask) {
#pragma unroll
                        for (int r = 0; r < NR; ++r) acc[mi][ni][r] *= om;
                    }
                    if (has_div) {
#pragma unroll
                        for (int r = 0; r < NR; ++r) acc[mi][ni][r] = acc[mi][ni][r] / out_div;
                    }
                }
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int rb = row0 + conv_erow<NR>(r, rq);
                    const int row = rb + 4 * h;
                    const bool rok = row < c_out;
                    float v = (MODE == TTSAMD_CONV_NORMAL) ? acc[mi][ni][r] : acc[mi][ni][r] + radd[r];
                    if constexpr (MODE == TTSAMD_CONV_SHUFFLE) {
                        const int u = shuffle_u;
                        if (shuffle_vec) {
                            // stride u % 4 == 0: a lane's four consecutive packed rows (r & 3 = 0..3) are four consecutive
                            // phases of one output channel = four consecutive output samples: ONE 16-byte store per group
                            // (4x fewer store instructions, 16 contiguous bytes per lane instead of 4 at a 4u-byte stride)
                            if ((r & 3) == 3) {
                                const int rowg = rb + 4 * h;                    // first row of the group (rb is row r&~3 ... + (r&3))
                                const int row_first = rowg - 3;
                                const int co = row_first / u;
                                const int n = t * u + (row_first - co * u) - shuffle_pad;
                                const bool ok = tv && (row_first + 3 < c_out) && n >= 0 && n + 3 < shuffle_t_out;
                                f32x4s q;
                                q[0] = acc[mi][ni][r - 3] + radd[r - 3];
                                q[1] = acc[mi][ni][r - 2] + radd[r - 2];
                                q[2] = acc[mi][ni][r - 1] + radd[r - 1];
                                q[3] = v;
                                if (ok) *reinterpret_cast<f32x4s *>(y + (long)b * y_bs + (long)co * y_rs + n) = q;
                            }
                        } else if (shuffle_vec2) {
                            // stride 2 (ups[2], ups[3]): packed rows (2c, 2c+1) are the two phases of channel c = two consecutive
                            // samples: one 8-byte store per pair (4-byte aligned: a global store; tile-edge columns fall back)
                            if (r & 1) {
                                const int co = (row - 1) >> 1;
                                const int n = t * 2 - 1;                        // sample of phase 0 (pad = 1)
                                const float v0 = acc[mi][ni][r - 1] + radd[r - 1];
                                if (tv && rok && n >= 0 && n + 1 < shuffle_t_out) {
                                    f32x2u q;
                                    q[0] = v0;
                                    q[1] = v;
                                    *reinterpret_cast<f32x2u *>(y + (long)b * y_bs + (long)co * y_rs + n) = q;
                                } else if (tv && rok) {
                                    if (n >= 0 && n < shuffle_t_out) y[(long)b * y_bs + (long)co * y_rs + n] = v0;
                                    if (n + 1 >= 0 && n + 1 < shuffle_t_out) y[(long)b * y_bs + (long)co * y_rs + n + 1] = v;
                                }
                            }
                        } else {
                            const int co = row / u;
                            const int rr = row - co * u;
                            const int n = t * u + rr - shuffle_pad;
                            const bool ok = tv && rok && n >= 0 && n < shuffle_t_out;
                            st_buf(ry, v, ok ? (co * y_rs4 + n * 4) : kOob, 0);
                        }
                    } else if constexpr (MODE == TTSAMD_CONV_COUPLE) {
                        v = v * om;
                        v = (e1[r] - v) * om;
                        st_buf(ry, v, rok ? voy : kOob, rb * y_rs4);
                    } else if constexpr (MODE == TTSAMD_CONV_RES_SKIP) {
                        if (lower) {
                            v = (e1[r] + v) * om;
                            st_buf(ry, v, rok ? voy : kOob, rb * y_rs4);
                        } else {
                            v = e2[r] + v;
                            st_buf(ry2, v, rok ? voy2 : kOob, (rb - split) * y2_rs4);
                        }
                    } else {
                        // NORMAL: finished above.  c_out a multiple of 32 (wave-uniform): every row of the tile exists, no compare +
                        // select per store
                        st_buf(ry, v, (all_rows || rok) ? voy : kOob, rb * y_rs4);
                    }
                }
            }
        }
    }
}

template <int K, int D, int MI, int NI, int WM, int WN, int MODE>
__global__ __launch_bounds__(64 * WM * WN, (MI * NI >= 4) ? 3 : 4) void conv1d_mfma_kernel(const ttsamd_conv1d_args a)
{
    using G = ConvGeom<K, D, MI, NI, WM, WN>;
    static_assert(((kConvCK / 2) * K) % 4 == 0, "k-steps per chunk must be a multiple of 4");
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [2][CK][XW]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR): row offsets become scalar soffsets, no waterfall loops
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int h = lane >> 5;   // which channel of the pair / which row group of D
    const int j = lane & 31;   // column inside a 32-wide N tile
    const ConvTile tile = conv_tile_of_block();
    const int b = tile.b;
    const int mb = tile.mb;
    const int t0 = tile.nb * G::kBN;
    const int nchunks = (a.c_in + kConvCK - 1) / kConvCK;
    const long ksg_total = (long)nchunks * G::kGroupsPerChunk;  // groups per m-tile

    // ---- global memory goes through buffer resources (SRSRC + 32-bit offsets): one VGPR offset per access, no
    // 64-bit address arithmetic, and out-of-range lanes are dropped / read as 0 by the hardware range check
    // (invalid lanes get kOob as their offset).  One resource per (tensor, batch item) slab; slabs are < 2 GiB
    // (checked on the host).
    constexpr int kOob = kConvOob;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)b * a.x_bstride, ((long)(a.c_in - 1) * a.x_rstride + a.t_in) * 4);

    // Staged element i of this thread: row = channel inside the chunk, col = column of the [BN + halo] strip.
    // Everything but the chunk's channel offset is chunk-independent and computed ONCE: byte offset (kOob when the
    // column is outside [0, t_in) or the slot is padding) and the input-mask value.
    int soff[G::kNStage];
    float smask[G::kNStage];
#pragma unroll
    for (int i = 0; i < G::kNStage; ++i) {
        const int e = tid + i * G::kThreads;
        const int row = e / G::kXW;
        const int col = e - row * G::kXW;
        const int gt = t0 - a.pad_left + col;
        const bool ok = (e < G::kStageElems) && (gt >= 0) && (gt < a.t_in);
        soff[i] = ok ? (int)(((long)row * a.x_rstride + gt) * 4) : kOob;
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
    const int chunk_bytes = kConvCK * (int)a.x_rstride * 4;   // channel offset of one chunk (slab < 2 GiB)
    float st[G::kNStage];
    auto stage_load = [&](int chunk) {
        const int cb = chunk * chunk_bytes;
#pragma unroll
        for (int i = 0; i < G::kNStage; ++i)   // channels >= c_in fall outside the slab -> 0
            st[i] = ld_buf(rx, soff[i] == kOob ? kOob : soff[i] + cb, 0);
    };
    auto stage_store = [&](float *buf) {
#pragma unroll
        for (int i = 0; i < G::kNStage; ++i) {
            const int e = tid + i * G::kThreads;
            if (e < G::kStageElems) buf[e] = conv_in_act(st[i] * smask[i], a.in_act, a.in_slope);
        }
    };

    // ---- accumulators ----------------------------------------------------------------------
    f32x16 acc[MI][NI];

    // A fragment stream of this wave: m-tile (blockIdx.y*WM + wm)*MI + mi
    const float4 *wp[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const long mtile = ((long)mb * WM + wm) * MI + mi;
        wp[mi] = reinterpret_cast<const float4 *>(a.w_packed) + (mtile * ksg_total) * 64 + lane;
    }
    float4 a_cur[MI], a_nxt[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a_cur[mi] = wp[mi][0];

    stage_load(0);
    const bool folded = conv_acc_init<MODE, MI, NI, WM, WN>(acc, a, b, mb, t0, wm, wn, h, j);
    // materialise the accumulators in AGPRs here: the residual loads' temporaries must not stay live in the loop
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) asm volatile("" : "+a"(acc[mi][ni]));
    stage_store(xs);
    __syncthreads();

    const int bcol = h * G::kXW + wn * (32 * NI) + j;  // base LDS index of this lane's B reads
    for (int c = 0; c < nchunks; ++c) {
        const float *cur = xs + (c & 1) * G::kStageElems;
        if (c + 1 < nchunks) stage_load(c + 1);
        const float *bbase = cur + bcol;
        constexpr int kSteps = G::kGroupsPerChunk * 4;
        // B fragments are register double-buffered one k-step ahead of the MFMAs that use them.
        float bf[2][NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bf[0][ni] = bbase[ni * 32];
#pragma unroll
        for (int gl = 0; gl < G::kGroupsPerChunk; ++gl) {
            const long g = (long)c * G::kGroupsPerChunk + gl;
            // prefetch the next group's A fragments (the packed image has one zero group of slack)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a_nxt[mi] = wp[mi][(g + 1) * 64];
            // pin the prefetch a full group (4 k-steps of MFMAs) ahead of its first use; hipcc
            // otherwise sinks the loads down to their consumer and exposes the L2 latency.
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int ks = gl * 4 + s;       // compile-time after unrolling
                if (ks + 1 < kSteps) {
                    const int p1 = (ks + 1) / K;
                    const int tap1 = (ks + 1) - p1 * K;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        bf[(ks + 1) & 1][ni] = bbase[(2 * p1) * G::kXW + ni * 32 + tap1 * D];
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const float av = (s == 0) ? a_cur[mi].x : (s == 1) ? a_cur[mi].y : (s == 2) ? a_cur[mi].z : a_cur[mi].w;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[ks & 1][ni], acc[mi][ni], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a_cur[mi] = a_nxt[mi];
        }
        if (c + 1 < nchunks) stage_store(xs + ((c + 1) & 1) * G::kStageElems);
        __syncthreads();
    }

    conv_epilogue<MODE, MI, NI, WM, WN>(acc, b, mb, t0, wm, wn, h, j, folded);
}

template <int K, int D, int MI, int NI, int WM, int WN, int MODE>
int conv1d_launch_cfg(const ttsamd_conv1d_args &a, hipStream_t st)
{
    using G = ConvGeom<K, D, MI, NI, WM, WN>;
    auto kern = conv1d_mfma_kernel<K, D, MI, NI, WM, WN, MODE>;
    static std::atomic<unsigned long long> lds_attr_done{0};   // per device, see ensure_dynamic_lds
    TTSAMD_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), (int)G::kLdsBytes, lds_attr_done));
    const int mtiles = (a.c_out + 31) / 32;
    const int mblocks = (mtiles + MI * WM - 1) / (MI * WM);
    const int nblocks = (a.t_out + G::kBN - 1) / G::kBN;
    hipLaunchKernelGGL(kern, dim3(nblocks, mblocks, a.batch), dim3(G::kThreads), G::kLdsBytes, st, a);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

// Tile choice by packed row count: 128x128 (4 waves, 2x2 of 64x64), 64x256, 32x256.
template <int K, int D, int MODE>
int conv1d_launch_tiles(const ttsamd_conv1d_args &a, hipStream_t st)
{
    const int mtiles = (a.c_out + 31) / 32;
    // (the paired-row modes pair inside a 32-row tile since round 4: they tile like every other mode)
    if (mtiles % 4 == 0) return conv1d_launch_cfg<K, D, 2, 2, 2, 2, MODE>(a, st);
    if (mtiles % 2 == 0) return conv1d_launch_cfg<K, D, 2, 2, 1, 4, MODE>(a, st);
    return conv1d_launch_cfg<K, D, 1, 2, 1, 4, MODE>(a, st);
}

}  // namespace ttsamd
