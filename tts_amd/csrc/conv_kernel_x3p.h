// Persistent, epilogue-pipelined variant of the split-bf16 conv (conv_kernel_x3.h) for TTSAMD_CONV_NORMAL.
//
// Why: with one block per output tile, every block runs [residual read] -> [MFMA main loop] -> [64 store instructions
// per wave], and because all blocks of a launch take the same time they stay in phase: the whole chip alternates between
// an HBM-bound phase (matrix pipes idle) and an MFMA-bound phase (HBM idle).  Measured with per-phase clock stamps
// (scripts/phase_clocks.py, 128-channel k=3 layer): prologue 12 k + epilogue 23 k cycles against a 44 k-cycle main loop.
//
// Here a block is persistent (grid = CUs x 2) and walks a strided list of tiles.  The finished tile's accumulators move to
// a second register set and its epilogue is executed in SLICES (one 32x32 tile = 16 registers per lane) inside the next
// tile's main loop: slice p's residual is requested at the start of channel chunk p (together with the chunk's staging
// loads), combined after the first tap, and stored — fire and forget — while the remaining taps run.  The next tile's
// first activation chunk and first weight fragments are prefetched during the current tile's last chunk, so the chunk
// pipeline (LDS double buffer, one barrier per chunk) never drains at a tile boundary.  HBM traffic is thereby spread
// evenly under the MFMA stream instead of bracketing it.
//
// Geometry: a wave owns MI x NI 32x32 tiles with NI = 4 where possible: 3 weight loads (16 B/lane, L2) feed 24 MFMAs,
// the activation fragments come from LDS (4x the L1 bandwidth), so the L1/L2 path carries half the bytes per MFMA of the
// 2x2 arrangement.  The bias initialises the accumulators (every tile of a block has the same m-block: the grid is a
// multiple of the m-block count), so the epilogue needs no per-row operand.
//
// Arithmetic is the split-bf16 scheme of conv_kernel_x3.h, unchanged.
//
// RESULT (MI355X, B=32 VITS-decoder shapes): isolated launches x1.03-1.07 on the 128-row k=3 layers, +-1 % on k=7/11,
// x0.82-0.96 on the 64/32-row layers; end to end 95.1 (off) / 95.2 (128-row layers) / 97.9 ms (everywhere).  The phases
// overlap as designed, but the chip is POWER-limited on this kernel family (shader clock 1.55-1.9 GHz of 2.4 with real
// operands, 2.4 GHz with all-zero operands: scripts/ubench/x3_tiles.hip c): the joules per tile are unchanged, so
// overlapping the HBM phase with the MFMA phase only lowers the clock both run at.  Kept selectable
// (ttsamd_conv1d_set_pipeline) and covered by tests/test_conv_gpu.py; OFF by default.
#pragma once
#include "conv_kernel_x3.h"

namespace ttsamd {

extern int g_conv_pipeline;   // conv.hip: 0 = off (default), 1 = 128-row m-blocks, 2 = every eligible NORMAL launch

template <int K, int D, int MI, int NI, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv1d_x3p_kernel(const ttsamd_conv1d_args a, int nblk_n, int mblocks, int ntiles)
{
    using G = ConvGeomX3<K, D, MI, NI, WM, WN>;
    constexpr int kSlices = MI * NI;
    constexpr int kOob = kConvOob;
    extern __shared__ __attribute__((aligned(16))) unsigned char xs3[];  // [2][3 parts][XW][16 ch] bf16

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: row offsets become SGPR soffsets
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int h = lane >> 5;
    const int j = lane & 31;
    const int nchunks = (a.c_in + kConvCK - 1) / kConvCK;

    // Logical block id: consecutive ids sit on one XCD (the hardware deals blockIdx round-robin over the 8 XCDs), so
    // the m-blocks / neighbouring time tiles that share an activation tile share an L2.
    int lid = blockIdx.x;
    if ((gridDim.x & 7) == 0) lid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int mb = lid % mblocks;   // constant per block: gridDim.x is a multiple of mblocks
    auto decode = [&](int tile, int &b, int &t0) {
        const int q = tile / mblocks;
        b = q / nblk_n;
        t0 = (q - b * nblk_n) * G::kBN;
    };

    // ---- staging context (the tile whose activations are being fetched) ------------------------------------------
    int soff[G::kNStage];
    float smask[G::kNStage];
    __amdgpu_buffer_rsrc_t rx;
    auto set_stage_ctx = [&](int b, int t0) {
        rx = make_rsrc(a.x + (long)b * a.x_bstride, ((long)(a.c_in - 1) * a.x_rstride + a.t_in) * 4);
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
    };
    const int row_bytes = (int)a.x_rstride * 4;
    float st[G::kNStage][8];
    auto stage_load = [&](int chunk) {
        const int cb = chunk * kConvCK * row_bytes;
#pragma unroll
        for (int i = 0; i < G::kNStage; ++i)
#pragma unroll
            for (int c = 0; c < 8; ++c) st[i][c] = ld_buf(rx, soff[i] == kOob ? kOob : soff[i] + cb + c * row_bytes, 0);
    };
    auto stage_store = [&](unsigned char *buf) {
#pragma unroll
        for (int i = 0; i < G::kNStage; ++i) {
            const int e = tid + i * G::kThreads;
            const int half = e / G::kXW;
            const int col = e - half * G::kXW;
            if (e < G::kItems) {
                unsigned p[3][8];
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    conv_split3(conv_in_act(st[i][c] * smask[i], a.in_act, a.in_slope), p[0][c], p[1][c], p[2][c]);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    u32x4 w;
                    w.x = p[q][0] | (p[q][1] << 16);
                    w.y = p[q][2] | (p[q][3] << 16);
                    w.z = p[q][4] | (p[q][5] << 16);
                    w.w = p[q][6] | (p[q][7] << 16);
                    *reinterpret_cast<u32x4 *>(buf + q * G::kPartBytes + col * 32 + half * 16) = w;
                }
            }
        }
    };

    // ---- weights: the A stream of this wave is the same for every tile of the block ---------------------------------
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
        for (int q = 0; q < 3; ++q) a_cur[mi][q] = wp[mi][q * 64];

    // The accumulators start from the bias of this lane's 16 rows (same rows for every tile of the block; re-read from
    // L1/L2 at each tile boundary rather than held in 16 registers).
    auto acc_init = [&](f32x16 (&acc)[MI][NI]) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row0 = ((mb * WM + wm) * MI + mi) * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                acc[mi][0][r] = (a.bias && row < a.c_out) ? a.bias[row] : 0.f;
            }
#pragma unroll
            for (int ni = 1; ni < NI; ++ni) acc[mi][ni] = acc[mi][0];
        }
    };

    int tile_c = lid, b_c, t0_c;
    decode(tile_c, b_c, t0_c);
    set_stage_ctx(b_c, t0_c);
    stage_load(0);

    f32x16 acc[MI][NI], prev[MI][NI];
    acc_init(acc);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) prev[mi][ni][r] = 0.f;
    stage_store(xs3);
    __syncthreads();

    // ---- deferred epilogue of the previous tile, one 32x32 slice at a time -------------------------------------------
    bool have_prev = false;
    int b_p = 0, t0_p = 0;
    float e[16];   // optional operand of the slice in flight (residual, then MRF accumulator)
    float om = 1.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) e[r] = 0.f;
    const ttsamd_conv1d_args __attribute__((address_space(4))) *ep =
        (const ttsamd_conv1d_args __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();

    // P0: request the residual (and the output mask) of slice (mi, ni)
    auto slice_p0 = [&](int mi, int ni) {
        const int t = t0_p + wn * (32 * NI) + ni * 32 + j;
        const bool tv = t < ep->t_out;
        const float *omask = ep->out_mask;
        om = omask ? omask[(long)b_p * ep->t_out + (tv ? t : 0)] : 1.f;
        const float *res = ep->res;
        if (res) {
            const int c_out = ep->c_out;
            const int rs4 = (int)ep->res_rstride * 4;
            const __amdgpu_buffer_rsrc_t rr = make_rsrc(res + (long)b_p * ep->res_bstride, ((long)(c_out - 1) * ep->res_rstride + ep->t_out) * 4);
            const int row0 = ((mb * WM + wm) * MI + mi) * 32;
            const int vo = tv ? (4 * h * rs4 + t * 4) : kOob;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rb = row0 + (r & 3) + 8 * (r >> 2);
                e[r] = ld_buf(rr, (rb + 4 * h < c_out) ? vo : kOob, rb * rs4);
            }
        }
    };
    // P1: + residual; request the MRF accumulator
    auto slice_p1 = [&](f32x16 &v, int mi, int ni) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += e[r];
        const float *accum = ep->accum;
        if (accum) {
            const int t = t0_p + wn * (32 * NI) + ni * 32 + j;
            const bool tv = t < ep->t_out;
            const int c_out = ep->c_out;
            const int rs4 = (int)ep->accum_rstride * 4;
            const __amdgpu_buffer_rsrc_t rr = make_rsrc(accum + (long)b_p * ep->accum_bstride, ((long)(c_out - 1) * ep->accum_rstride + ep->t_out) * 4);
            const int row0 = ((mb * WM + wm) * MI + mi) * 32;
            const int vo = tv ? (4 * h * rs4 + t * 4) : kOob;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rb = row0 + (r & 3) + 8 * (r >> 2);
                e[r] = ld_buf(rr, (rb + 4 * h < c_out) ? vo : kOob, rb * rs4);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = 0.f;
        }
    };
    // P2: + accumulator, mask, store
    auto slice_p2 = [&](f32x16 &v, int mi, int ni) {
        const int t = t0_p + wn * (32 * NI) + ni * 32 + j;
        const bool tv = t < ep->t_out;
        const int c_out = ep->c_out;
        const int rs4 = (int)ep->y_rstride * 4;
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(ep->y + (long)b_p * ep->y_bstride, ((long)(c_out - 1) * ep->y_rstride + ep->t_out) * 4);
        const int row0 = ((mb * WM + wm) * MI + mi) * 32;
        const int vo = tv ? (4 * h * rs4 + t * 4) : kOob;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rb = row0 + (r & 3) + 8 * (r >> 2);
            st_buf(ry, (e[r] + v[r]) * om, (rb + 4 * h < c_out) ? vo : kOob, rb * rs4);
            e[r] = 0.f;
        }
    };

    const int bbyte = (wn * (32 * NI) + j) * 32 + h * 16;   // this lane's fragment inside a part, tap 0, ni 0
    int par = 0;
    for (;;) {
        const int tile_n = tile_c + (int)gridDim.x;
        const bool has_next = tile_n < ntiles;
        for (int c = 0; c < nchunks; ++c) {
            const unsigned char *cur = xs3 + par * G::kBufBytes + bbyte;
            const bool last = (c + 1 == nchunks);
            bool staged = true;
            if (!last) {
                stage_load(c + 1);
            } else if (has_next) {
                int b_n, t0_n;
                decode(tile_n, b_n, t0_n);
                set_stage_ctx(b_n, t0_n);
                stage_load(0);
            } else {
                staged = false;
            }
            const bool do_slice = have_prev && c < kSlices;
            if (do_slice) {
#pragma unroll
                for (int p = 0; p < kSlices; ++p)
                    if (c == p) slice_p0(p / NI, p % NI);
            }
#pragma unroll
            for (int tap = 0; tap < K; ++tap) {
                // next tap's weights; after the last tap of a tile: the first tap of the next tile (same stream)
                const long g = (tap == K - 1 && last) ? 0 : ((long)c * K + tap + 1) * (3 * 64);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int q = 0; q < 3; ++q) a_nxt[mi][q] = wp[mi][g + q * 64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    u32x4 bq[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        bq[q] = *reinterpret_cast<const u32x4 *>(cur + q * G::kPartBytes + (ni * 32 + tap * D) * 32);
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
                if (tap == 0 && do_slice) {
#pragma unroll
                    for (int p = 0; p < kSlices; ++p)
                        if (c == p) slice_p1(prev[p / NI][p % NI], p / NI, p % NI);
                }
            }
            if (do_slice) {
#pragma unroll
                for (int p = 0; p < kSlices; ++p)
                    if (c == p) slice_p2(prev[p / NI][p % NI], p / NI, p % NI);
            }
            if (staged) stage_store(xs3 + (par ^ 1) * G::kBufBytes);
            __syncthreads();
            par ^= 1;
        }
        // slices the main loop had no chunk for (c_in < 16 * slices)
        if (have_prev && nchunks < kSlices) {
#pragma unroll
            for (int p = 0; p < kSlices; ++p)
                if (p >= nchunks) {
                    slice_p0(p / NI, p % NI);
                    slice_p1(prev[p / NI][p % NI], p / NI, p % NI);
                    slice_p2(prev[p / NI][p % NI], p / NI, p % NI);
                }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) prev[mi][ni] = acc[mi][ni];
        acc_init(acc);
        have_prev = true;
        b_p = b_c;
        t0_p = t0_c;
        if (!has_next) break;
        tile_c = tile_n;
        decode(tile_c, b_c, t0_c);
    }
    // the last tile's epilogue
#pragma unroll
    for (int p = 0; p < kSlices; ++p) {
        slice_p0(p / NI, p % NI);
        slice_p1(prev[p / NI][p % NI], p / NI, p % NI);
        slice_p2(prev[p / NI][p % NI], p / NI, p % NI);
    }
}

template <int K, int D, int MI, int NI, int WM, int WN>
int conv1d_x3p_launch_cfg(const ttsamd_conv1d_args &a, hipStream_t st)
{
    using G = ConvGeomX3<K, D, MI, NI, WM, WN>;
    auto kern = conv1d_x3p_kernel<K, D, MI, NI, WM, WN>;
    static int slots = 0;   // resident blocks on the whole device
    if (!slots) {
        TTSAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)G::kLdsBytes));
        int dev = 0, cus = 0, occ = 0;
        TTSAMD_HIP(hipGetDevice(&dev));
        TTSAMD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        TTSAMD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(kern), G::kThreads, G::kLdsBytes));
        slots = cus * (occ > 0 ? occ : 1);
    }
    const int mtiles = (a.c_out + 31) / 32;
    const int mblocks = (mtiles + MI * WM - 1) / (MI * WM);
    const int nblk_n = (a.t_out + G::kBN - 1) / G::kBN;
    const long ntiles_l = (long)nblk_n * mblocks * a.batch;
    if (ntiles_l > 0x7fffffffL) {
        set_error("conv1d: too many tiles");
        return TTSAMD_ERR_UNSUPPORTED;
    }
    const int ntiles = (int)ntiles_l;
    int grid = ntiles < slots ? ntiles : slots;
    grid -= grid % mblocks;            // every block keeps one m-block (ntiles is a multiple of mblocks)
    if (grid >= 8 * mblocks) grid -= grid % (8 * mblocks);   // XCD-contiguous logical ids
    hipLaunchKernelGGL(kern, dim3(grid), dim3(G::kThreads), G::kLdsBytes, st, a, nblk_n, mblocks, ntiles);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

// NORMAL-mode launches the pipelined kernel takes: no per-item row bias (the bias lives in the accumulators' initial
// value, shared by all tiles of a block), no output activation and no division (conv_post's tanh and the MRF mean's
// divide stay on the one-block-per-tile kernel: their register-hungry epilogues would push this kernel into spills).
inline bool conv1d_x3p_eligible(const ttsamd_conv1d_args &a)
{
    // measured (scripts/pipe_ab.py, B=32 VITS-decoder shapes): x1.05-1.07 on the 128-row k=3 layers, +-1 % on k=7/11,
    // x0.82-0.96 on the 64- and 32-row layers (their tiles are HBM-bound either way) -> 128-row m-blocks only
    return a.row_bias == nullptr && a.out_act == TTSAMD_ACT_NONE && a.out_div == 0.f && (((a.c_out + 31) / 32) % 4 == 0 || g_conv_pipeline > 1);
}

template <int K, int D>
int conv1d_x3p_launch_tiles(const ttsamd_conv1d_args &a, hipStream_t st)
{
    const int mtiles = (a.c_out + 31) / 32;
    if (mtiles % 4 == 0) return conv1d_x3p_launch_cfg<K, D, 1, 4, 4, 1>(a, st);
    if (mtiles % 2 == 0) return conv1d_x3p_launch_cfg<K, D, 1, 4, 2, 2>(a, st);
    return conv1d_x3p_launch_cfg<K, D, 1, 2, 1, 4>(a, st);
}

}  // namespace ttsamd
