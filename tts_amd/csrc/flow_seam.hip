// The seam between two Glow-TTS flow blocks of a single sentence as ONE launch (round 4):
//
//     out  = (out + res_skip_last(acts) + b) * mask                   last WaveNet layer's 1x1 conv      wavenet.py:109-115
//     t|s  = end(out) + b                                             CouplingBlock.end, 1x1              glow.py:214-215
//     x1   = (x1 - t) * exp(-s) * mask                                affine coupling, reverse            glow.py:221-224
//     x    = ActNorm^-1(InvConvNear^-1(x))                            4x4 channel mix + per-channel affine glow.py:107-137
//     h'   = start'(x0) * mask                                        NEXT block's CouplingBlock.start    glow.py:199-201
//
// All five are pointwise in time, and at a sentence's ~160 squeezed frames each was a launch of a few dozen blocks whose
// time is its own latency chain (~5 us each, 4 launches per flow block, 12 blocks per sentence).  Here a block owns 32
// columns and ALL channels: the three 1x1 convs run back to back on the matrix pipe with their operands handed over in LDS
// (split-bf16 planes, the layout of conv_kernel_x3.h), 12 waves = (m-tile 0..5) x (K half 0..1); the two K halves of a
// tile meet in LDS in a fixed order.  Same arithmetic per product as the conv kernels; the reduction is cut into two
// slices per tile instead of the one-shot kernel's twelve (fp32 re-association): tests compare at 2e-6 relative.
#include "conv_kernel_x3.h"

namespace ttsamd {

using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

constexpr int kSeamH = 192;                 // WaveNet hidden channels (GlowTTSConfig.hidden_channels_dec)
constexpr int kSeamNCH = kSeamH / 16;       // 12 chunks
constexpr int kSeamMT = kSeamH / 32;        // 6 m-tiles
constexpr int kSeamPlane = 32 * 16;         // bytes of one [32 columns][8 ch] bf16 plane
constexpr int kSeamThreads = 64 * 12;

__device__ __forceinline__ unsigned char *seam_plane(unsigned char *base, int nch, int q, int chunk, int half)
{
    return base + ((q * nch + chunk) * 2 + half) * kSeamPlane;
}

// acc += sum over chunks [c0, c0 + n) of W[m-tile][chunk] x B[chunk]: weights of all n chunks requested up front
template <int N>
__device__ __forceinline__ void seam_gemm(f32x16 &acc, const u32x4 *wp, int nch_img, int c0, int n, unsigned char *bbase, int nch_lds, int h,
                                          int j)
{
    u32x4 aw[N][3];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int c = (i < n) ? c0 + i : c0;        // (a short slice re-reads its first chunk; the product is skipped below)
#pragma unroll
        for (int q = 0; q < 3; ++q) aw[i][q] = wp[(long)c * (3 * 64) + q * 64];
    }
    (void)nch_img;
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (i < n) {
            u32x4 bq[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) bq[q] = *reinterpret_cast<const u32x4 *>(seam_plane(bbase, nch_lds, q, c0 + i, h) + j * 16);
            constexpr int pa[6] = {2, 1, 0, 1, 0, 0};
            constexpr int pb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                if (t & 1)
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[i][pa[t]]),
                                                                   __builtin_bit_cast(bf16x8, bq[pb[t]]), acc2, 0, 0, 0);
                else
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[i][pa[t]]),
                                                                  __builtin_bit_cast(bf16x8, bq[pb[t]]), acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
}

// four consecutive channels (idx 4h .. 4h+3 of an 8-channel half) of one column -> three bf16 parts -> LDS
__device__ __forceinline__ void seam_store4(unsigned char *base, int nch, int chunk, int half, int h, int j, float v0, float v1, float v2,
                                            float v3)
{
    unsigned pw[3][2];
    conv_split3x2(v0, v1, pw[0][0], pw[1][0], pw[2][0]);
    conv_split3x2(v2, v3, pw[0][1], pw[1][1], pw[2][1]);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        u32x2 w;
        w.x = pw[q][0];
        w.y = pw[q][1];
        *reinterpret_cast<u32x2 *>(seam_plane(base, nch, q, chunk, half) + j * 16 + h * 8) = w;
    }
}

__global__ __launch_bounds__(kSeamThreads) void glow_flow_seam_kernel(const ttsamd_flow_seam_args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int half_c = a.half;                               // coupled channels (80)
    const int nch_x0 = (half_c + 15) / 16;                   // chunks of the next block's start conv (5)
    unsigned char *const ldsA = lds;                                           // acts planes      [3][12][2][32 x 16 B]
    unsigned char *const ldsB = ldsA + 3 * kSeamNCH * 2 * kSeamPlane;          // out planes       [3][12][2]
    unsigned char *const ldsC = ldsB + 3 * kSeamNCH * 2 * kSeamPlane;          // x0' planes       [3][8][2]  (up to 128 channels)
    float *const red = reinterpret_cast<float *>(ldsC + 3 * 8 * 2 * kSeamPlane);   // [6 tiles][16][64] partial tiles of K half 1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = wave % kSeamMT;
    const int kh = wave / kSeamMT;
    const int h = lane >> 5;
    const int j = lane & 31;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * 32;
    const int T = a.t;
    const int t = t0 + j;
    const bool tv = t < T;
    constexpr int kOob = kConvOob;
    const int C = 2 * half_c;

    const float mk = a.mask ? (tv ? a.mask[(long)b * T + t] : 0.f) : (tv ? 1.f : 0.f);

    // ---- phase 0: stage acts: wave w = chunk w (16 channels x 32 columns); lane = (8-channel half, column) -------------
    {
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(a.acts + (long)b * kSeamH * T, (long)kSeamH * T * 4);
        const int off = tv ? (int)(((long)(wave * 16 + 8 * h) * T + t) * 4) : kOob;
        float st[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) st[i] = ld_buf(ra, off == kOob ? kOob : off + i * T * 4, 0);
        unsigned pw[3][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) conv_split3x2(st[2 * i], st[2 * i + 1], pw[0][i], pw[1][i], pw[2][i]);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            u32x4 w;
            w.x = pw[q][0];
            w.y = pw[q][1];
            w.z = pw[q][2];
            w.w = pw[q][3];
            *reinterpret_cast<u32x4 *>(seam_plane(ldsA, kSeamNCH, q, wave, h) + j * 16) = w;
        }
    }
    // operands of the epilogues, requested now: skip accumulator rows of tile m, the coupling / mixing operands of tile m
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(a.out + (long)b * kSeamH * T, (long)kSeamH * T * 4);
    float e_out[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m * 32 + (r & 3) + 8 * (r >> 2);
        e_out[r] = (kh == 0 && a.accumulate) ? ld_buf(ro, tv ? (4 * h * T + t) * 4 : kOob, row * T * 4) : 0.f;
    }
    float *const xb = a.x + (long)b * C * T;
    float e_x0[8], e_x1[8];
    bool pok[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;          // channel of the pair tile: rows i (t) and 16 + i (s)
        const int oc = m * 16 + i;
        pok[r] = tv && kh == 0 && m * 32 + 16 + i < a.end_rows && oc < half_c;
        e_x0[r] = pok[r] ? xb[(long)oc * T + t] : 0.f;
        e_x1[r] = pok[r] ? xb[(long)(half_c + oc) * T + t] : 0.f;
    }
    __syncthreads();

    // ---- phase 1: out = (out + W_rs acts + b) * mask -> LDS planes ------------------------------------------------------
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
        const u32x4 *wp = reinterpret_cast<const u32x4 *>(a.w_rs) + (long)m * (kSeamNCH * 3 * 64) + lane;
        seam_gemm<6>(acc, wp, kSeamNCH, kh * 6, 6, ldsA, kSeamNCH, h, j);
    }
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(m * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[(m * 16 + r) * 64 + lane];
        float o[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = acc[r] + (a.b_rs ? a.b_rs[row] : 0.f);
            v = e_out[r] + v;
            o[r] = v * mk;
        }
        if (a.out_store) {                                   // the caller wants the WaveNet output too (parity tests)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (tv) a.out_store[((long)b * kSeamH + row) * T + t] = o[r];
            }
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
            seam_store4(ldsB, kSeamNCH, 2 * m + (rg >> 1), rg & 1, h, j, o[rg * 4], o[rg * 4 + 1], o[rg * 4 + 2], o[rg * 4 + 3]);
    }
    __syncthreads();

    // ---- phase 2: t|s = W_end out + b; coupling; InvConvNear^-1; ActNorm^-1 -> x (global) and x0' (LDS planes) -------------
    const int end_tiles = (a.end_rows + 31) / 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (m < end_tiles) {
        const u32x4 *wp = reinterpret_cast<const u32x4 *>(a.w_end) + (long)m * (kSeamNCH * 3 * 64) + lane;
        seam_gemm<6>(acc, wp, kSeamNCH, kh * 6, 6, ldsB, kSeamNCH, h, j);
    }
    if (kh == 1 && m < end_tiles) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(m * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kh == 0 && m < end_tiles) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[(m * 16 + r) * 64 + lane];
        const float *const mixp = a.mix;                     // [16] 4x4 inverse, [C] bias, [C] logs
        float w[4][4];
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4)
#pragma unroll
            for (int b4 = 0; b4 < 4; ++b4) w[a4][b4] = mixp[a4 * 4 + b4];
        float x0n[8];
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int oc = m * 16 + i;
            const int prow = m * 32 + i;
            const bool ok = pok[r] && pok[r + 1];
            float v[4];
            v[0] = e_x0[r];
            v[1] = e_x0[r + 1];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float tt = acc[r + u] + (a.b_end ? a.b_end[ok ? prow + u : 0] : 0.f);
                const float ss = acc[r + u + 8] + (a.b_end ? a.b_end[ok ? prow + u + 16 : 0] : 0.f);
                v[2 + u] = (e_x1[r + u] - tt) * expf(-ss) * mk;
            }
            float z4[4];
#pragma unroll
            for (int go = 0; go < 4; ++go) {
                float z = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) z += w[go][g] * v[g];
                const int ch = (go >> 1) * half_c + oc + (go & 1);
                z *= mk;
                z = (z - mixp[16 + (ok ? ch : 0)]) * expf(-mixp[16 + C + (ok ? ch : 0)]) * mk;
                z4[go] = ok ? z : 0.f;
                if (ok) xb[(long)ch * T + t] = z;
            }
            x0n[r] = z4[0];
            x0n[r + 1] = z4[1];
        }
        // x0' channels m*16 + {0..3, 8..11} + 4h -> chunk m, half (r >> 2), idx 4h..4h+3
        seam_store4(ldsC, 8, m, 0, h, j, x0n[0], x0n[1], x0n[2], x0n[3]);
        seam_store4(ldsC, 8, m, 1, h, j, x0n[4], x0n[5], x0n[6], x0n[7]);
    }
    if (!a.w_start) return;                                  // the last block of the stack: no next start conv
    __syncthreads();

    // ---- phase 3: h' = (W_start' x0' + b) * mask ------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
        const int c0 = kh == 0 ? 0 : (nch_x0 + 1) / 2;
        const int n = kh == 0 ? (nch_x0 + 1) / 2 : nch_x0 - (nch_x0 + 1) / 2;
        const u32x4 *wp = reinterpret_cast<const u32x4 *>(a.w_start) + (long)m * ((long)nch_x0 * 3 * 64) + lane;
        seam_gemm<4>(acc, wp, nch_x0, c0, n, ldsC, 8, h, j);
    }
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(m * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float v = (acc[r] + red[(m * 16 + r) * 64 + lane]) + (a.b_start ? a.b_start[row] : 0.f);
            if (tv) a.h_next[((long)b * kSeamH + row) * T + t] = v * mk;
        }
    }
}

}  // namespace ttsamd
using namespace ttsamd;

extern "C" int ttsamd_glow_flow_seam(const ttsamd_flow_seam_args *args, void *stream)
{
    TTSAMD_CHECK_ARG(args, "glow_flow_seam: NULL args");
    const ttsamd_flow_seam_args &a = *args;
    TTSAMD_CHECK_ARG(a.acts && a.out && a.x && a.w_rs && a.w_end && a.mix, "glow_flow_seam: NULL tensor");
    TTSAMD_CHECK_ARG(a.hidden == kSeamH, "glow_flow_seam: hidden channels %d (built for %d)", a.hidden, kSeamH);
    TTSAMD_CHECK_ARG(a.half > 0 && a.half % 2 == 0 && a.half <= 96 && a.end_rows == 32 * ((a.half + 15) / 16),
                     "glow_flow_seam: coupled channels %d (even, <= 96) / packed end rows %d", a.half, a.end_rows);
    TTSAMD_CHECK_ARG(!a.w_start || a.h_next, "glow_flow_seam: w_start without h_next");
    TTSAMD_CHECK_ARG(a.t >= 0 && a.batch >= 0 && a.batch <= 65535, "glow_flow_seam: bad shape");
    if (a.t == 0 || a.batch == 0) return TTSAMD_OK;
    constexpr size_t kLds = (size_t)(2 * 3 * kSeamNCH * 2 + 3 * 8 * 2) * kSeamPlane + (size_t)kSeamMT * 16 * 64 * 4;
    static_assert(kLds <= 160 * 1024, "flow seam: LDS budget");
    static std::atomic<unsigned long long> lds_attr_done{0};
    TTSAMD_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(glow_flow_seam_kernel), (int)kLds, lds_attr_done));
    hipLaunchKernelGGL(glow_flow_seam_kernel, dim3((a.t + 31) / 32, a.batch), dim3(kSeamThreads), kLds, as_stream(stream), a);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}
