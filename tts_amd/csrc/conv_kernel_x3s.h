// Split-bf16 Conv1d for SMALL GRIDS (a single utterance: text encoder, duration predictors, flow layers — launches of a
// few dozen 64-column tiles on a 256-CU chip).  Same arithmetic, weight image, LDS image and epilogues as
// conv_kernel_x3.h; what differs is what the block time is made of.  scripts/phase_clocks.py on the general kernel's
// small-grid tiles (192 -> 192, k = 1, T = 257): 18 300 cycles per block for 18 MFMAs per wave — 3 000 prologue, 9 100 in a
// three-iteration K loop, 6 100 epilogue: chains of exposed memory round trips (each K iteration waits for its activation
// chunk, each tap for its weight fragments, the epilogue for ten dependent scalar loads).  And a 64x64 tile with four
// wave groups is ONE workgroup = ONE CU: 768 -> 192, k = 3 spends 4 650 cycles per K iteration whatever the prefetch
// depth, because 16 waves' staging VALU work (~2 100 cycles per SIMD) and their MFMAs (~2 300) take turns between the
// iteration's barriers — 15 CUs of 256 busy.
//
// Here a block is KS wave groups of WM x WN waves (one 32x32 tile per wave and m-tile: 32*MI*WM rows x 32*WN columns per
// block; 1 x 1 — a wave per tile and K slice, 4x the blocks of the 64x64 tile — unless that would launch > 1024 blocks);
// group g owns the channel chunks (it*KS + g)*CPI .. +CPI-1 of K-loop iteration `it`:
//   * weights: one register slot per (chunk-in-iteration, tap); a slot is re-requested for the NEXT iteration as soon as
//     its MFMAs are issued, so every fragment has a whole iteration (CPI*K tap steps + the staging work) to arrive;
//   * activations: CPI chunks (16*CPI channels) are requested, split and staged per iteration; at C_in = 192 and k = 1
//     (CPI = 3, KS = 4) the whole reduction is ONE iteration — all loads of the launch are in flight together;
//   * staging items are (column, 4-channel quarter), spread evenly over the group's lanes;
//   * a single-wave group alternates two accumulators (its six products per tap would otherwise be one dependent chain);
//   * partial tiles of groups 1..KS-1 meet in LDS (aliased onto the dead activation buffers) in a fixed order.
#pragma once
#include "conv_kernel_x3.h"

namespace ttsamd {

using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

template <int K, int D, int MI, int WM, int WN, int CPI>
struct ConvGeomX3S {
    static constexpr int kGroupThreads = 64 * WM * WN;
    static constexpr int kBN = 32 * WN;
    static constexpr int kHalo = (K - 1) * D;
    static constexpr int kXW = kBN + kHalo;                  // staged columns
    static constexpr int kPartBytes = kXW * 32;              // [half][column][8 ch] bf16
    static constexpr int kChunkBytes = 3 * kPartBytes;
    static constexpr int kBufBytes = CPI * kChunkBytes;      // one iteration of one group
    static constexpr int kItems = 4 * kXW;                   // (column, 4-channel quarter) items per chunk
    static constexpr int kRounds = (kItems + kGroupThreads - 1) / kGroupThreads;
    static constexpr bool kPartial = (kItems % kGroupThreads) != 0;
    static constexpr int kTileFloats = MI * 16 * 64;         // one wave's accumulators
};

template <int K, int D, int MI, int WM, int WN, int MODE, int KS, int CPI, bool ONE>   // ONE: the whole reduction is a single iteration
__global__ __launch_bounds__(64 * WM * WN * KS, (KS * WM * WN >= 4 ? KS * WM * WN / 4 : 1)) void conv1d_x3s_kernel(const ttsamd_conv1d_args a)
{
    using G = ConvGeomX3S<K, D, MI, WM, WN, CPI>;
    constexpr int GT = G::kGroupThreads;
    constexpr bool kDual = (WM * WN == 1);
    extern __shared__ __attribute__((aligned(16))) unsigned char xs_all[];
    const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x / GT);
    const int tid = (int)threadIdx.x - grp * GT;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int h = lane >> 5;
    const int j = lane & 31;
    const ConvTile tile = conv_tile_of_block();
    const int b = tile.b;
    const int mb = tile.mb;
    const int t0 = tile.nb * G::kBN;
    const int nchunks = (a.c_in + kConvCK - 1) / kConvCK;
    const int niter = (nchunks + KS * CPI - 1) / (KS * CPI);
    unsigned char *const xs = xs_all + (size_t)grp * (ONE ? 1 : 2) * G::kBufBytes;
#ifdef TTSAMD_PHASE_CLOCKS
    long long pc[5];
    pc[0] = clock64();
    const long long prt0 = wall_clock64();
#endif

    constexpr int kOob = kConvOob;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)b * a.x_bstride, ((long)(a.c_in - 1) * a.x_rstride + a.t_in) * 4);
    const int row_bytes = (int)a.x_rstride * 4;

    // staging item of round r: (quarter, column) = (e / XW, e % XW), e = tid + GT r
    int soff[G::kRounds], loff[G::kRounds];
    float smask[G::kRounds];
#pragma unroll
    for (int r = 0; r < G::kRounds; ++r) {
        const int e = tid + r * GT;
        const int q4 = e / G::kXW;
        const int col = e - q4 * G::kXW;
        const int gt = t0 - a.pad_left + col;
        const bool ok = (e < G::kItems) && (gt >= 0) && (gt < a.t_in);
        soff[r] = ok ? (int)(((long)(q4 * 4) * a.x_rstride + gt) * 4) : kOob;
        loff[r] = (q4 >> 1) * (G::kXW * 16) + col * 16 + (q4 & 1) * 8;
        smask[r] = 1.f;
    }
    if (a.in_mask) {
        const __amdgpu_buffer_rsrc_t rm = make_rsrc(a.in_mask + (long)b * a.t_in, (long)a.t_in * 4);
#pragma unroll
        for (int r = 0; r < G::kRounds; ++r) {
            const int e = tid + r * GT;
            const int col = e - (e / G::kXW) * G::kXW;
            const int gt = t0 - a.pad_left + col;
            smask[r] = ld_buf(rm, (e < G::kItems && gt >= 0 && gt < a.t_in) ? gt * 4 : kOob, 0);
        }
    }
    float st[CPI][G::kRounds][4];
    auto stage_load = [&](int it) {
#pragma unroll
        for (int cc = 0; cc < CPI; ++cc) {
            const int cb = ((it * KS + grp) * CPI + cc) * kConvCK * row_bytes;   // chunks beyond c_in read as zeros (range check)
#pragma unroll
            for (int r = 0; r < G::kRounds; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) st[cc][r][c] = ld_buf(rx, soff[r] == kOob ? kOob : soff[r] + cb + c * row_bytes, 0);
        }
    };
    // the last round is partial when 4*XW is not a multiple of the group's lanes (k > 1: the halo columns); its idle lanes
    // split zeros and write them to a private 8-byte dump slot instead of sitting behind an exec-mask branch (see below)
    unsigned char *const dump = xs_all + (size_t)KS * (ONE ? 1 : 2) * G::kBufBytes + (size_t)threadIdx.x * 8;   // after every group's buffers
    const bool last_valid = tid + (G::kRounds - 1) * GT < G::kItems;
    auto stage_store = [&](unsigned char *buf) {
#pragma unroll
        for (int cc = 0; cc < CPI; ++cc)
#pragma unroll
            for (int r = 0; r < G::kRounds; ++r) {
#if TTSAMD_SPLIT_PAIRS
                unsigned pw[3][2];
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    conv_split3x2(conv_in_act(st[cc][r][2 * c] * smask[r], a.in_act, a.in_slope),
                                  conv_in_act(st[cc][r][2 * c + 1] * smask[r], a.in_act, a.in_slope), pw[0][c], pw[1][c], pw[2][c]);
#else
                unsigned p[3][4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    conv_split3(conv_in_act(st[cc][r][c] * smask[r], a.in_act, a.in_slope), p[0][c], p[1][c], p[2][c]);
#endif
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    u32x2 w;
#if TTSAMD_SPLIT_PAIRS
                    w.x = pw[q][0];
                    w.y = pw[q][1];
#else
                    w.x = p[q][0] | (p[q][1] << 16);
                    w.y = p[q][2] | (p[q][3] << 16);
#endif
                    unsigned char *dst = buf + cc * G::kChunkBytes + q * G::kPartBytes + loff[r];
                    if (G::kPartial && r == G::kRounds - 1) dst = last_valid ? dst : dump;
                    *reinterpret_cast<u32x2 *>(dst) = w;
                }
            }
    };

    // weight slots of this wave: m-tile (mb*WM + wm)*MI + mi, image [chunk][tap][part][64 lanes] x 16 bytes
    const u32x4 *wp[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const long mtile = ((long)mb * WM + wm) * MI + mi;
        wp[mi] = reinterpret_cast<const u32x4 *>(a.w_split) + mtile * ((long)nchunks * K * 3 * 64) + lane;
    }
    // Every vector-memory request below is issued UNCONDITIONALLY (indices clamped, never branched around): s_waitcnt
    // counts are exact only on straight-line code — behind a wave-uniform `if` the compiler has to assume the younger
    // requests may not exist and drains the whole queue (vmcnt(0)) before every use, which serialises the slot ring.
    // A group's chunks beyond the input (its last iteration) stage zeros (buffer range check) against the clamped, finite
    // weights of the last chunk: they add +0.
    u32x4 aw[CPI * K][MI][3];
    auto weights_load = [&](int slot, int chunk) {
        const long cl = chunk < nchunks ? chunk : nchunks - 1;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int q = 0; q < 3; ++q) aw[slot][mi][q] = wp[mi][(cl * K + slot % K) * (3 * 64) + q * 64];
    };

    stage_load(0);
#pragma unroll
    for (int sl = 0; sl < CPI * K; ++sl) weights_load(sl, grp * CPI + sl / K);
    f32x16 acc[MI][1];
    f32x16 acc2[kDual ? MI : 1];
#pragma unroll
    for (int mi = 0; mi < (kDual ? MI : 1); ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[mi][r] = 0.f;
    bool folded = false;
    if (grp == 0) {
        folded = conv_acc_init<MODE, MI, 1, WM, WN>(acc, a, b, mb, t0, wm, wn, h, j);
    } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][0][r] = 0.f;
    }
    stage_store(xs);
    // drain the prologue's requests (weights, folded residual) HERE: the first MFMA needs them anyway, and a loop whose
    // entry state has requests in flight gets the entry's conservative wait count on every iteration
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
#ifdef TTSAMD_PHASE_CLOCKS
    pc[1] = clock64();
#endif

    const int bbyte = h * (G::kXW * 16) + (wn * 32 + j) * 16;   // this lane's fragment inside a part, tap 0
    auto tap_steps = [&](const unsigned char *cur, int it) {
#pragma unroll
        for (int sl = 0; sl < CPI * K; ++sl) {
            constexpr int kTapBytes = D * 16;
            const int cc = sl / K, tap = sl % K;
            u32x4 bq[3];
#pragma unroll
            for (int q = 0; q < 3; ++q)
                bq[q] = *reinterpret_cast<const u32x4 *>(cur + cc * G::kChunkBytes + q * G::kPartBytes + tap * kTapBytes);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                constexpr int pa[6] = {2, 1, 0, 1, 0, 0};   // smallest products first (as conv1d_x3_kernel)
                constexpr int pb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    if (kDual && (t & 1))
                        acc2[kDual ? mi : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, aw[sl][mi][pa[t]]), __builtin_bit_cast(bf16x8, bq[pb[t]]), acc2[kDual ? mi : 0], 0, 0, 0);
                    else
                        acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, aw[sl][mi][pa[t]]), __builtin_bit_cast(bf16x8, bq[pb[t]]), acc[mi][0], 0, 0, 0);
                }
            }
            // this slot's fragment of the next iteration: a whole iteration ahead of its use
            if constexpr (!ONE) {
                weights_load(sl, ((it + 1) * KS + grp) * CPI + cc);
                __builtin_amdgcn_sched_barrier(0);   // keep the request here, a whole iteration ahead of its use
            }
        }
    };
    if constexpr (ONE) {
        tap_steps(xs + bbyte, 0);
        __syncthreads();                          // the partial tiles below overwrite the activation image
    } else {
        // straight-line iterations (see the note at weights_load): the iteration after the last stages zeros into the idle
        // buffer — a few hundred cycles of split work, cheaper than what a branch around it does to the wait counts
        for (int it = 0; it < niter; ++it) {
            stage_load(it + 1);
            tap_steps(xs + (it & 1) * G::kBufBytes + bbyte, it);
            stage_store(xs + ((it + 1) & 1) * G::kBufBytes);
            __syncthreads();
        }
    }
    if constexpr (kDual) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][0][r] += acc2[mi][r];
    }

    if constexpr (KS > 1) {
        // partial tiles of groups 1..KS-1 -> LDS (over the activation buffers: every wave is past the loop's last barrier)
        // -> group 0, which adds them in group order and alone runs the epilogue
        float *red = reinterpret_cast<float *>(xs_all);
        if (grp > 0) {
            float *dst = red + ((size_t)(grp - 1) * (WM * WN) + wave) * G::kTileFloats + lane;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(mi * 16 + r) * 64] = acc[mi][0][r];
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll 1
        for (int g = 1; g < KS; ++g) {
            const float *src = red + ((size_t)(g - 1) * (WM * WN) + wave) * G::kTileFloats + lane;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][0][r] += src[(mi * 16 + r) * 64];
        }
    }
#ifdef TTSAMD_PHASE_CLOCKS
    pc[2] = clock64();
#endif
    conv_epilogue<MODE, MI, 1, WM, WN>(acc, b, mb, t0, wm, wn, h, j, folded);
#ifdef TTSAMD_PHASE_CLOCKS
    pc[3] = clock64();
    __builtin_amdgcn_s_waitcnt(0);
    pc[4] = clock64();
    if ((MODE == TTSAMD_CONV_NORMAL || MODE == TTSAMD_CONV_GATE) && a.y2 && threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == gridDim.z / 2) {
        long long *o = reinterpret_cast<long long *>(a.y2);
        for (int i = 0; i < 5; ++i) o[i] = pc[i] - pc[0];
        o[5] = wall_clock64() - prt0;
    }
#endif
}

// LDS of one launch: KS groups x (1 or 2) iteration buffers (+ the dump slots of a partial staging round), or the
// (KS - 1) x WM x WN partial tiles if those are larger
template <int K, int D, int MI, int WM, int WN, int KS, int CPI>
constexpr size_t conv1d_x3s_lds_bytes(bool one)
{
    using G = ConvGeomX3S<K, D, MI, WM, WN, CPI>;
    const size_t stage = (size_t)KS * (one ? 1 : 2) * G::kBufBytes + (G::kPartial ? (size_t)KS * G::kGroupThreads * 8 : 0);
    const size_t red = (size_t)(KS - 1) * (WM * WN) * G::kTileFloats * sizeof(float);
    return stage > red ? stage : red;
}

template <int K, int D, int MI, int WM, int WN, int MODE, int KS, int CPI, bool ONE>
int conv1d_x3s_launch_one(const ttsamd_conv1d_args &a, hipStream_t st)
{
    using G = ConvGeomX3S<K, D, MI, WM, WN, CPI>;
    constexpr size_t kLds = conv1d_x3s_lds_bytes<K, D, MI, WM, WN, KS, CPI>(ONE);
    static_assert(kLds <= 160 * 1024, "conv1d_x3s: LDS budget");
    auto kern = conv1d_x3s_kernel<K, D, MI, WM, WN, MODE, KS, CPI, ONE>;
    static std::atomic<unsigned long long> lds_attr_done{0};   // per device, see ensure_dynamic_lds
    TTSAMD_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), (int)kLds, lds_attr_done));
    const int mtiles = (a.c_out + 31) / 32;
    const int mblocks = (mtiles + WM * MI - 1) / (WM * MI);
    const int nblocks = (a.t_out + G::kBN - 1) / G::kBN;
    hipLaunchKernelGGL(kern, dim3(nblocks, mblocks, a.batch), dim3(G::kGroupThreads * KS), kLds, st, a);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

template <int K, int D, int MI, int WM, int WN, int MODE, int KS, int CPI>
int conv1d_x3s_launch_geom(const ttsamd_conv1d_args &a, hipStream_t st)
{
    const int nchunks = (a.c_in + kConvCK - 1) / kConvCK;
    if constexpr (K == 1) {   // the single-iteration form is only built where it occurs: 1x1 convs of <= 192 channels
        if (nchunks <= KS * CPI) return conv1d_x3s_launch_one<K, D, MI, WM, WN, MODE, KS, CPI, true>(a, st);
    }
    return conv1d_x3s_launch_one<K, D, MI, WM, WN, MODE, KS, CPI, false>(a, st);
}

// MI = 2: paired rows (GATE).  Returns false when the shape has no instantiation here (the caller falls through).
template <int K, int D, int MI, int MODE>
bool conv1d_x3s_launch(const ttsamd_conv1d_args &a, hipStream_t st, int *rc)
{
    constexpr int CPI = (K == 1) ? 3 : 1;
    const int mtiles = (a.c_out + 31) / 32;
    if (mtiles % MI) return false;
    const long blocks32 = (long)((a.t_out + 31) / 32) * (mtiles / MI) * a.batch;
    if (blocks32 <= kConvWaveTileBlocks) {
        // a wave per (32-row, 32-column) tile and K slice; eight slices when the reduction is long (k = 3 at >= 512 channels)
        if constexpr (K == 3 && MODE == TTSAMD_CONV_NORMAL) {
            if (a.c_in >= 32 * kConvCK) {
                // round 4: sixteen slices (FFN conv_2, 768 channels: three K iterations instead of six) while the launch is a
                // handful of tiles; eight beyond
                if (blocks32 <= 64) *rc = conv1d_x3s_launch_geom<K, D, MI, 1, 1, MODE, 16, CPI>(a, st);
                else *rc = conv1d_x3s_launch_geom<K, D, MI, 1, 1, MODE, 8, CPI>(a, st);
                return true;
            }
        }
        *rc = conv1d_x3s_launch_geom<K, D, MI, 1, 1, MODE, 4, CPI>(a, st);
        return true;
    }
    // longer launches (the waveform decoder's first stage of a single utterance): 64x64 tiles, four waves per group.  Paired
    // rows never get here with a small grid (their 32-column tiling has 8x the default blocks, the unpaired one 16x).
    if constexpr (MI == 1) {
        if (mtiles % 2 == 0) {
            // round 4: K >= 3 at >= 128 output rows (one utterance's 256-channel decoder stage: 256 -> 256, T = 6160) — the general
            // kernel's 128-row x 64-column blocks (four waves of 32 x 64: a weight fragment serves two column tiles, half the L1
            // bytes per MFMA of the 32 x 32 wave tiles below) with the K loop split between two wave groups.  Same box, us per
            // launch k = 3 / 7 / 11: 30.0 / 49.6 / 69.4 -> 24.3 / 39.0 / 56.2; the B = 1 request 4.27 -> 4.11 ms
            // (TTSAMD_X3S_WIDE=0 restores the 64 x 64 tiles for A/B runs)
            static const bool wide = [] { const char *e = getenv("TTSAMD_X3S_WIDE"); return !e || atoi(e) != 0; }();
            if constexpr (MODE == TTSAMD_CONV_NORMAL && K >= 3) {
                if (wide && mtiles % 4 == 0 && (long)(mtiles / 4) * ((a.t_out + 63) / 64) * a.batch >= 128) {   // else too few blocks (128 rows, T = 2624: 41)
                    *rc = conv1d_x3_launch_cfg<K, D, 1, 2, 4, 1, MODE, 2>(a, st);
                    return true;
                }
            }
            *rc = conv1d_x3s_launch_geom<K, D, 1, 2, 2, MODE, (K > 5 ? 2 : 4), CPI>(a, st);   // k = 7 / 11 (TTSAMD_X3S_ALL): K weight slots need > 128 VGPRs
            return true;
        }
    }
    return false;
}

}  // namespace ttsamd
