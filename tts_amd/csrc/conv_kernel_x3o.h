// Split-bf16 Conv1d for small grids, ONE-SHOT form (round 4): every K slice of a 32x32 output tile is its own wave, the whole
// reduction is requested up front, and the tile's epilogue is spread over four waves.
//
// What the looping small-grid kernel (conv_kernel_x3s.h) spends at the single-sentence shapes (scripts/phase_clocks.py, WaveNet
// gate conv 192 -> 384, k = 5, T = 159: 48 launches of a Glow-TTS sentence, 16 of a VITS request): 29 700 cycles per block =
// prologue 3 600 + three K iterations of 4 400 (1 920 of them MFMA: each iteration's weight fragments come from the far
// side of the chip — 2.2 MB of weights per layer, read once per request — and one iteration of lead does not cover that) +
// 12 200 epilogue (ONE wave reducing four partial tiles and evaluating 16 tanh x sigmoid per lane).  Here:
//   * a block is KS single-wave groups; wave g owns channel chunks g*CPI .. g*CPI + CPI - 1 (16 channels each) and ALL K taps
//     of them: KS * CPI >= c_in / 16, so there is no K loop — every activation and weight request of the launch is in flight
//     before the first MFMA (straight-line code: exact wait counts);
//   * a wave stages its own chunk(s) into its own LDS slice (fp32 -> three bf16 planes, exactly as conv_kernel_x3.h) and
//     reads only that slice: the arithmetic per product is that of the other split-bf16 kernels;
//   * the KS partial tiles meet in LDS; waves 0..3 each add them up — in wave order, a fixed summation order — for ONE
//     QUARTER of the tile's accumulator registers and run the fused epilogue on that quarter (conv_epilogue<.., NR = 4>):
//     a quarter of the transcendental work and of the store instructions per wave, four waves in parallel.
// Same weight image, LDS image and epilogue code as the other kernels; fp32 reassociation relative to them (K is cut into
// more slices), covered by the same tolerances (tests/test_conv_gpu.py::test_small_grid_tiles_and_k_split).
#pragma once
#include "conv_kernel_x3s.h"

namespace ttsamd {

template <int K, int D, int CPI>
struct ConvGeomX3O {
    static constexpr int kBN = 32;
    static constexpr int kHalo = (K - 1) * D;
    static constexpr int kXW = kBN + kHalo;                  // staged columns
    static constexpr int kPartBytes = kXW * 32;              // [half][column][8 ch] bf16
    static constexpr int kChunkBytes = 3 * kPartBytes;
    static constexpr int kBufBytes = CPI * kChunkBytes;      // one wave's slice
    static constexpr int kItems = 4 * kXW;                   // (column, 4-channel quarter) items per chunk
    static constexpr int kRounds = (kItems + 63) / 64;
    static constexpr bool kPartial = (kItems % 64) != 0;
};

template <int K, int D, int KS, int CPI>
constexpr size_t conv1d_x3o_lds_bytes()
{
    using G = ConvGeomX3O<K, D, CPI>;
    const size_t stage = (size_t)KS * G::kBufBytes + (G::kPartial ? (size_t)KS * 64 * 8 : 0);
    const size_t red = (size_t)KS * 16 * 64 * sizeof(float);
    return stage > red ? stage : red;
}

template <int K, int D, int MODE, int KS, int CPI>
__global__ __launch_bounds__(64 * KS) void conv1d_x3o_kernel(const ttsamd_conv1d_args a)
{
    using G = ConvGeomX3O<K, D, CPI>;
    static_assert(KS >= 4, "four waves share the epilogue");
    extern __shared__ __attribute__((aligned(16))) unsigned char xs_all[];
    const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = (int)threadIdx.x & 63;
    const int h = lane >> 5;
    const int j = lane & 31;
    const ConvTile tile = conv_tile_of_block();
    const int b = tile.b;
    const int mb = tile.mb;
    const int t0 = tile.nb * G::kBN;
    const int nchunks = (a.c_in + kConvCK - 1) / kConvCK;
    unsigned char *const xs = xs_all + (size_t)grp * G::kBufBytes;

    constexpr int kOob = kConvOob;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (long)b * a.x_bstride, ((long)(a.c_in - 1) * a.x_rstride + a.t_in) * 4);
    const int row_bytes = (int)a.x_rstride * 4;

    // staging item of round r: (quarter, column) = (e / XW, e % XW), e = lane + 64 r
    int soff[G::kRounds], loff[G::kRounds];
    float smask[G::kRounds];
#pragma unroll
    for (int r = 0; r < G::kRounds; ++r) {
        const int e = lane + r * 64;
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
            const int e = lane + r * 64;
            const int col = e - (e / G::kXW) * G::kXW;
            const int gt = t0 - a.pad_left + col;
            smask[r] = ld_buf(rm, (e < G::kItems && gt >= 0 && gt < a.t_in) ? gt * 4 : kOob, 0);
        }
    }
    // every request of the launch, unconditionally (chunks beyond c_in read zeros through the buffer range check against
    // the clamped, finite weights of the last chunk: they add +0)
    float st[CPI][G::kRounds][4];
#pragma unroll
    for (int cc = 0; cc < CPI; ++cc) {
        const int cb = (grp * CPI + cc) * kConvCK * row_bytes;
#pragma unroll
        for (int r = 0; r < G::kRounds; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) st[cc][r][c] = ld_buf(rx, soff[r] == kOob ? kOob : soff[r] + cb + c * row_bytes, 0);
    }
    const long mtile = mb;
    const u32x4 *const wp = reinterpret_cast<const u32x4 *>(a.w_split) + mtile * ((long)nchunks * K * 3 * 64) + lane;
    u32x4 aw[CPI * K][3];
#pragma unroll
    for (int sl = 0; sl < CPI * K; ++sl) {
        const int chunk = grp * CPI + sl / K;
        const long cl = chunk < nchunks ? chunk : nchunks - 1;
#pragma unroll
        for (int q = 0; q < 3; ++q) aw[sl][q] = wp[(cl * K + sl % K) * (3 * 64) + q * 64];
    }
    f32x16 acc[1][1];
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    // NORMAL mode without an output activation: wave 0's partial tile starts from the residual operand (conv_acc_init)
    const bool folded = (MODE == TTSAMD_CONV_NORMAL) && (a.out_act == TTSAMD_ACT_NONE) && a.res;
    if (grp == 0) {
        conv_acc_init<MODE, 1, 1, 1, 1>(acc, a, b, mb, t0, 0, 0, h, j);
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    }

    // split + stage this wave's chunk(s); the idle lanes of a partial last round write a private dump slot (no exec-mask branch)
    unsigned char *const dump = xs_all + (size_t)KS * G::kBufBytes + (size_t)threadIdx.x * 8;
    const bool last_valid = lane + (G::kRounds - 1) * 64 < G::kItems;
#pragma unroll
    for (int cc = 0; cc < CPI; ++cc)
#pragma unroll
        for (int r = 0; r < G::kRounds; ++r) {
            unsigned p[3][4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                conv_split3(conv_in_act(st[cc][r][c] * smask[r], a.in_act, a.in_slope), p[0][c], p[1][c], p[2][c]);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                u32x2 w;
                w.x = p[q][0] | (p[q][1] << 16);
                w.y = p[q][2] | (p[q][3] << 16);
                unsigned char *dst = xs + cc * G::kChunkBytes + q * G::kPartBytes + loff[r];
                if (G::kPartial && r == G::kRounds - 1) dst = last_valid ? dst : dump;
                *reinterpret_cast<u32x2 *>(dst) = w;
            }
        }
    __syncthreads();     // (a wave reads only its own slice; the barrier also orders the cross-lane LDS traffic inside it)

    const unsigned char *const cur = xs + h * (G::kXW * 16) + j * 16;   // this lane's fragment inside a part, tap 0
#pragma unroll
    for (int sl = 0; sl < CPI * K; ++sl) {
        constexpr int kTapBytes = D * 16;
        const int cc = sl / K, tap = sl % K;
        u32x4 bq[3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
            bq[q] = *reinterpret_cast<const u32x4 *>(cur + cc * G::kChunkBytes + q * G::kPartBytes + tap * kTapBytes);
        constexpr int pa[6] = {2, 1, 0, 1, 0, 0};   // smallest products first (as conv1d_x3_kernel)
        constexpr int pb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t) {               // two alternating accumulators: six products are otherwise one dependent chain
            if (t & 1)
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[sl][pa[t]]),
                                                               __builtin_bit_cast(bf16x8, bq[pb[t]]), acc2, 0, 0, 0);
            else
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[sl][pa[t]]),
                                                                    __builtin_bit_cast(bf16x8, bq[pb[t]]), acc[0][0], 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] += acc2[r];

    // partial tiles -> LDS (over the staging slices: every wave is past its last fragment read) -> waves 0..3, a quarter each
    float *const red = reinterpret_cast<float *>(xs_all);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((size_t)grp * 16 + r) * 64 + lane] = acc[0][0][r];
    __syncthreads();
    if constexpr (MODE == TTSAMD_CONV_SHUFFLE) {
        // the polyphase stores write a lane's four consecutive rows as one vector: the whole tile stays with wave 0
        if (grp >= 1) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = red[(size_t)r * 64 + lane];
#pragma unroll
            for (int g = 1; g < KS; ++g) s += red[((size_t)g * 16 + r) * 64 + lane];
            acc[0][0][r] = s;
        }
        conv_epilogue<MODE, 1, 1, 1, 1>(acc, b, mb, t0, 0, 0, h, j, folded);
        return;
    }
    if (grp >= 4) return;
    f32x16 q4[1][1];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = 2 * grp + (rr & 1) + 8 * (rr >> 1);          // registers {2q, 2q+1, 2q+8, 2q+9}: conv_erow<4>
        float s = red[(size_t)r * 64 + lane];
#pragma unroll
        for (int g = 1; g < KS; ++g) s += red[((size_t)g * 16 + r) * 64 + lane];
        q4[0][0][rr] = s;
    }
    if constexpr (MODE != TTSAMD_CONV_SHUFFLE) conv_epilogue<MODE, 1, 1, 1, 1, 4>(q4, b, mb, t0, 0, 0, h, j, folded, grp);
}

template <int K, int D, int MODE, int KS, int CPI>
int conv1d_x3o_launch_one(const ttsamd_conv1d_args &a, hipStream_t st)
{
    constexpr size_t kLds = conv1d_x3o_lds_bytes<K, D, KS, CPI>();
    static_assert(kLds <= 160 * 1024, "conv1d_x3o: LDS budget");
    auto kern = conv1d_x3o_kernel<K, D, MODE, KS, CPI>;
    static std::atomic<unsigned long long> lds_attr_done{0};   // per device, see ensure_dynamic_lds
    TTSAMD_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), (int)kLds, lds_attr_done));
    const int mtiles = (a.c_out + 31) / 32;
    const int nblocks = (a.t_out + 31) / 32;
    hipLaunchKernelGGL(kern, dim3(nblocks, mtiles, a.batch), dim3(64 * KS), kLds, st, a);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

constexpr long kConvOneShotBlocks = 640;   // up to this many 32x32 tiles (a 12-wave block per tile) a launch takes the one-shot kernel

// Returns false when the shape has no instantiation here (the caller falls through to the other small-grid kernels).
template <int K, int D, int MODE>
bool conv1d_x3o_launch(const ttsamd_conv1d_args &a, hipStream_t st, int *rc)
{
    if constexpr (D != 1 || K > 7 || (MODE == TTSAMD_CONV_SHUFFLE) != (K == 2)) {
        return false;
    } else {
        const int mtiles = (a.c_out + 31) / 32;
        const int nchunks = (a.c_in + kConvCK - 1) / kConvCK;
        const long blocks = (long)((a.t_out + 31) / 32) * mtiles * a.batch;
        // up to 640 twelve-wave blocks (a 32x32 tile each); lighter blocks (fewer K slices) in proportion
        const int ks_est = (K == 1) ? (nchunks <= 12 ? 4 : (nchunks <= 24 ? 8 : 16)) : (nchunks <= 4 ? 4 : (nchunks <= 8 ? 8 : (nchunks <= 12 ? 12 : 16)));
        if (blocks * ks_est > kConvOneShotBlocks * 12 || a.t_out < 1) return false;
        // chunks per wave: three for 1x1 convs (36 weight registers), else one
        if constexpr (K == 1) {
            if (nchunks <= 12) *rc = conv1d_x3o_launch_one<K, D, MODE, 4, 3>(a, st);
            else if (nchunks <= 24) *rc = conv1d_x3o_launch_one<K, D, MODE, 8, 3>(a, st);
            else if (nchunks <= 48) *rc = conv1d_x3o_launch_one<K, D, MODE, 16, 3>(a, st);
            else return false;
            return true;
        } else {
            // one chunk per wave; k >= 5 keeps K * 12 weight registers per wave: up to 12 waves (168 registers each), k = 3 up to 16
            if (nchunks <= 4) *rc = conv1d_x3o_launch_one<K, D, MODE, 4, 1>(a, st);
            else if (nchunks <= 8) *rc = conv1d_x3o_launch_one<K, D, MODE, 8, 1>(a, st);
            else if (K == 2) return false;      // (the whole-tile polyphase epilogue spills beyond eight waves' register budget)
            else if (nchunks <= 12) *rc = conv1d_x3o_launch_one<K, D, MODE, (K == 2 ? 8 : 12), 1>(a, st);
            else {
                if constexpr (K == 3) {
                    if (nchunks <= 16) {
                        *rc = conv1d_x3o_launch_one<K, D, MODE, 16, 1>(a, st);
                        return true;
                    }
                }
                return false;       // longer reductions (FFN conv_2: 768 channels) keep the looping kernel
            }
            return true;
        }
    }
}

}  // namespace ttsamd
