// Relative-position attention, small-grid form (round 4: integrated from the round-2/3 study scripts/ubench/att_v2.hip).
// The block of tts_amd/csrc/attention.hip re-cut for the phase times measured in round 2
// (profiles/r02_attention_phase_clocks.txt, 89 k cycles per 32-query block at T = 257: QK^T 27 k, softmax 22.5 k, P.V 30 k):
//   * 8 waves per block instead of 4: the key tiles of QK^T go one per wave (the next tile's K fragment is requested
//     before the current tile's 48 dependent fp32 MFMAs), P.V is split over (channel tile, key-tile subset) pairs and the
//     2-3 partial tiles of a channel tile meet in LDS in a fixed order;
//   * softmax keeps a wave's 4 rows in REGISTERS between its passes (one LDS read + one write per score instead of three
//     of each, and no read-after-write chains through LDS);
//   * P fragments are ds_read_b128 (row pitch = 4 mod 32 floats: conflict free), V fragments 16-byte global loads.
// Same arithmetic as rel_attention_kernel (exact fp32 MFMA products, scores divided by sqrt(dk) after the contraction, -1e4
// mask fill, expf) — only summation orders inside P.V change.  39.8 -> 27.7 us at B = 1, T = 257 (six launches of a VITS
// request), +-0 at B = 32: taken for launches of up to kAttV2MaxBlocks blocks.  Included by attention.hip only.
#pragma once

namespace att2 {
using namespace ttsamd;

constexpr int kRows = 32;
constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr int kMaxSteps = 16;   // 64-column steps of a row: T <= 1024
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x4u = __attribute__((ext_vector_type(4), aligned(4))) float;

template <int DK>  // dk rounded up to a multiple of 32; channels dk..DK-1 are treated as zeros
__global__ __launch_bounds__(kThreads) void rel_attention_v2_kernel(
    float *__restrict__ out, const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    long qkv_bstride, const float *__restrict__ mask, const float *__restrict__ emb_k, const float *__restrict__ emb_v,
    int window, int heads, int dk, int T, int pitch)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NM = DK / 32;                 // channel tiles of the output
    const int ntiles = (T + 31) / 32;
    const int nrel = emb_k ? 2 * window + 1 : 0;
    float *S = smem;                            // [32][pitch] scores / probabilities, pitch % 4 == 0
    float *Ms = S + kRows * pitch;              // [ntiles*32] key mask (1 where absent)
    float *EkL = Ms + ntiles * 32;              // [nrel][DK] relative-key table, zero padded to DK
    float *EvL = EkL + nrel * DK;               // [nrel][DK] relative-value table
    float *Op = EvL + nrel * DK;                // [kWaves][16][64] partial output tiles
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5;
    const int j = lane & 31;
    const int t0 = blockIdx.x * kRows;
    const int head = blockIdx.y;
    const int b = blockIdx.z;
    const long hoff = (long)b * qkv_bstride + (long)head * dk * T;
    const float *qh = q + hoff, *kh = k + hoff, *vh = v + hoff;
    const float *mrow = mask ? mask + (long)b * T : nullptr;
    const float scale = sqrtf((float)dk);

    // ---- 0. small operands into LDS ----------------------------------------------------------------------------------
    for (int c = tid; c < ntiles * 32; c += kThreads) Ms[c] = (mrow && c < T) ? mrow[c] : 1.f;
    for (int e = tid; e < nrel * DK; e += kThreads) {
        const int r = e / DK, c = e - r * DK;
        EkL[e] = (c < dk) ? emb_k[r * dk + c] : 0.f;
        EvL[e] = (c < dk) ? emb_v[r * dk + c] : 0.f;
    }

    // ---- 1. S = Q K^T / sqrt(dk): one key tile per wave and round --------------------------------------------------
    const int slab = dk * T * 4;
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(qh, slab), rk = make_rsrc(kh, slab), rv = make_rsrc(vh, slab);
    float aq[DK / 2];
    {
        const bool qv = (t0 + j) < T;
#pragma unroll
        for (int ks = 0; ks < DK / 2; ++ks) {
            const int ch = 2 * ks + hh;
            aq[ks] = ld_buf(rq, (qv && ch < dk) ? (ch * T + t0 + j) * 4 : kBufOob, 0);
        }
    }
    float bk[DK <= 96 ? 2 : 1][DK / 2];
    // requests are unconditional (a tile past the end reads the out-of-range offset = 0, no memory traffic): wait counts
    // stay exact on straight-line code
    auto kload = [&](int jt, float(&dst)[DK / 2]) {
        const int col = jt * 32 + j;
        const bool kv = jt < ntiles && col < T;
#pragma unroll
        for (int ks = 0; ks < DK / 2; ++ks) {
            const int ch = 2 * ks + hh;
            dst[ks] = ld_buf(rk, (kv && ch < dk) ? (ch * T + col) * 4 : kBufOob, 0);
        }
    };
    auto ktile = [&](int jt, const float(&frag)[DK / 2]) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < DK / 2; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[ks], frag[ks], acc, 0, 0, 0);
        const int col = jt * 32 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
            S[row * pitch + col] = acc[r] / scale;
        }
    };
    if constexpr (DK <= 96) {
        kload(wave, bk[0]);
        for (int jt = wave; jt < ntiles; jt += 2 * kWaves) {
            kload(jt + kWaves, bk[1]);
            ktile(jt, bk[0]);
            kload(jt + 2 * kWaves, bk[0]);
            if (jt + kWaves < ntiles) ktile(jt + kWaves, bk[1]);
        }
    } else {   // dk > 96: two K fragments + the Q fragment do not fit 256 registers; more than 8 key tiles are T > 256 only
        for (int jt = wave; jt < ntiles; jt += kWaves) {
            kload(jt, bk[0]);
            ktile(jt, bk[0]);
        }
    }
    __syncthreads();

    // ---- 2a. relative-key band: S[i][i+d] += (Q[i] . Ek[d+w]) / sqrt(dk),  |d| <= w --------------------------------
    for (int r = wave; r < nrel; r += kWaves) {
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < DK / 2; ++ks) part += aq[ks] * EkL[r * DK + 2 * ks + hh];
        const float dot = part + __shfl_xor(part, 32);
        const int ti = t0 + j;
        const int tj = ti + r - window;
        if (hh == 0 && ti < T && tj >= 0 && tj < T) S[j * pitch + tj] += dot / scale;
    }
    if (nrel) __syncthreads();

    // ---- 2b. mask fill + softmax: a wave owns 4 rows and keeps them in registers between the passes ---------------
    {
        constexpr int RW = kRows / kWaves;   // 4
        float *Sw = S + wave * RW * pitch;
        const int nsteps = (T + 63) / 64;
        float mi[RW], mx[RW], sum[RW];
        float sv[RW][kMaxSteps];
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            mi[rr] = Ms[t0 + wave * RW + rr];
            mx[rr] = -INFINITY;
            sum[rr] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < kMaxSteps; ++s) {
            if (s < nsteps) {
                const int c = lane + 64 * s;
                const bool cv = c < T;
                const float mc = Ms[cv ? c : 0];
#pragma unroll
                for (int rr = 0; rr < RW; ++rr) {
                    float x = cv ? Sw[rr * pitch + c] : -INFINITY;
                    if (cv && mrow && (mi[rr] == 0.f || mc == 0.f)) x = -1e4f;
                    sv[rr][s] = x;
                    mx[rr] = fmaxf(mx[rr], x);
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) mx[rr] = wave_max(mx[rr]);
#pragma unroll
        for (int s = 0; s < kMaxSteps; ++s) {
            if (s < nsteps) {
#pragma unroll
                for (int rr = 0; rr < RW; ++rr) {
                    const float ev = (lane + 64 * s < T) ? expf(sv[rr][s] - mx[rr]) : 0.f;
                    sv[rr][s] = ev;
                    sum[rr] += ev;
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) sum[rr] = wave_sum(sum[rr]);
#pragma unroll
        for (int s = 0; s < kMaxSteps; ++s) {
            if (s < nsteps) {
                const int c = lane + 64 * s;
                if (c < ntiles * 32) {
#pragma unroll
                    for (int rr = 0; rr < RW; ++rr) Sw[rr * pitch + c] = (c < T) ? sv[rr][s] / sum[rr] : 0.f;
                }
            }
        }
    }
    __syncthreads();

    // ---- 3. O^T[n][i] = sum_kk V^T[n][kk] P^T[kk][i]: wave -> (channel tile m, every nw-th key tile) ------------------
    const int m = wave % NM;
    const int ksub = wave / NM;
    const int nw = (kWaves - m + NM - 1) / NM;          // waves sharing channel tile m
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
        const int n = m * 32 + j;                        // channel of this lane's V^T row (rows >= dk are zeros)
        const int nc = n < dk ? n : dk - 1;
        const float *vrow = vh + (long)nc * T;
        float vv[2][16];
        // full tiles: four unaligned 16-byte global loads per lane; the last, partial tile: range-checked dword loads
        auto vload = [&](int kt, float(&dst)[16]) {
            const int col0 = kt * 32 + 16 * hh;
            if (kt < ntiles && (kt + 1) * 32 <= T) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4u x = *reinterpret_cast<const f32x4u *>(vrow + col0 + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[4 * g + e] = x[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    dst[e] = ld_buf(rv, (kt < ntiles && col0 + e < T) ? (nc * T + col0 + e) * 4 : kBufOob, 0);
            }
            if (n >= dk) {
#pragma unroll
                for (int e = 0; e < 16; ++e) dst[e] = 0.f;
            }
        };
        auto ptile = [&](int kt, const float(&frag)[16]) {
            const float *prow = S + j * pitch + kt * 32 + 16 * hh;
            f32x4 p4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) p4[g] = *reinterpret_cast<const f32x4 *>(prow + 4 * g);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(frag[ks], p4[ks >> 2][ks & 3], acc, 0, 0, 0);
        };
        vload(ksub, vv[0]);
        for (int kt = ksub; kt < ntiles; kt += 2 * nw) {
            vload(kt + nw, vv[1]);
            ptile(kt, vv[0]);
            vload(kt + 2 * nw, vv[0]);
            if (kt + nw < ntiles) ptile(kt + nw, vv[1]);
        }
    }
    // partial tiles of the waves that share a channel tile -> LDS -> the first of them (ksub == 0), fixed order
    if (ksub > 0) {
        float *dst = Op + wave * 1024 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r * 64] = acc[r];
    }
    __syncthreads();
    if (ksub > 0) return;
    for (int w2 = m + NM; w2 < kWaves; w2 += NM) {
        const float *src = Op + w2 * 1024 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += src[r * 64];
    }

    // ---- 4. relative-value band + store -------------------------------------------------------------------------------
    const int ti = t0 + j;
    if (ti < T) {
        float rel[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rel[r] = 0.f;
        for (int d = 0; d < nrel; ++d) {
            const int tj = ti + d - window;
            const float p = (tj >= 0 && tj < T) ? S[j * pitch + tj] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) rel[r] += p * EvL[d * DK + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (n < dk) out[((long)b * heads * dk + (long)head * dk + n) * T + ti] = acc[r] + rel[r];
        }
    }
}

template <int DK>
int launch(float *out, const float *q, const float *k, const float *v, long bstride, const float *mask, const float *ek,
           const float *ev, int window, int batch, int heads, int dk, int T, hipStream_t st)
{
    const int ntiles = (T + 31) / 32;
    const int pitch = ntiles * 32 + 4;
    const int nrel = ek ? 2 * window + 1 : 0;
    const size_t lds = (size_t)(kRows * pitch + ntiles * 32 + 2 * nrel * DK + kWaves * 1024) * sizeof(float);
    if (lds > 160 * 1024) return -1;
    auto kern = rel_attention_v2_kernel<DK>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return -2;
    hipLaunchKernelGGL(kern, dim3(ntiles, heads, batch), dim3(kThreads), lds, st, out, q, k, v, bstride, mask, ek, ev, window, heads,
                       dk, T, pitch);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace att2
