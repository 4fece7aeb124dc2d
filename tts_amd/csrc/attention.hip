// Relative-position multi-head attention core for gfx950 (see include/tts_amd.h: ttsamd_rel_attention).
//
// Replaces RelativePositionMultiHeadAttention.attention (glow_tts/transformer.py:118-163) and its
// pad/reshape "skew" helpers (:196-241).  The reference multiplies Q by a zero-padded [2T-1, dk]
// table and re-indexes it by padding tricks; only the 2*window+1 diagonals |i-j| <= window are
// non-zero, so here they are added as a band directly.
//
// One workgroup (4 wavefronts) = 32 query rows of one (batch, head):
//   1. S = Q K^T / sqrt(dk)        fp32-input MFMA 32x32x2 (exact fp32 products).  Tensors are
//      channels-first [C, T], i.e. both operands are contiguous along their NON-contracted index,
//      which is exactly the MFMA fragment lane order -> A and B fragments are coalesced 128-byte
//      global loads, no LDS staging.  Scores land in an LDS strip S[32][T] (odd pitch).
//   2. band of relative-key logits, mask fill (-1e4), row softmax (wavefront shuffles).
//   3. O^T = V^T P^T               MFMA again; the contracted index is paired with (k-step, half-wave) such that a
//      lane's V values are consecutive in memory: V^T fragments come straight from global memory, P^T fragments from
//      the LDS strip.  Computing O^T (channels x time) makes the output stores coalesced along time.
//   4. band of relative-value terms added in registers, store.
#include "common.h"

#include <cstdlib>

namespace ttsamd {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kAttRows = 32;
constexpr int kAttThreads = 256;

#ifdef TTSAMD_PHASE_CLOCKS
__device__ long long g_att_clk[8];   // debug build: shader-clock stamps of one block (scripts/att_phase.py)
#define ATT_STAMP(i) do { if (tid == 0 && blockIdx.x == 1 && blockIdx.y == 0 && blockIdx.z == 0) g_att_clk[i] = clock64(); } while (0)
#else
#define ATT_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int DK>  // dk rounded up to a multiple of 32; channels dk..DK-1 are treated as zeros
__global__ __launch_bounds__(kAttThreads, (DK <= 96 ? 3 : 2)) void rel_attention_kernel(
    float *__restrict__ out, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, long qkv_bstride, const float *__restrict__ mask,
    const float *__restrict__ emb_k, const float *__restrict__ emb_v, int window, int heads, int dk, int T, int pitch)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ntiles = (T + 31) / 32;
    const int nrel = emb_k ? 2 * window + 1 : 0;
    float *S = smem;                            // [32][pitch] score / probability strip
    float *Ms = smem + kAttRows * pitch;        // [ntiles*32] key mask (1 where absent)
    float *EkL = Ms + ntiles * 32;              // [nrel][DK] relative-key table, zero padded to DK
    float *EvL = EkL + nrel * DK;               // [nrel][DK] relative-value table
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5;
    const int j = lane & 31;
    const int t0 = blockIdx.x * kAttRows;
    const int head = blockIdx.y;
    const int b = blockIdx.z;
    const long hoff = (long)b * qkv_bstride + (long)head * dk * T;
    const float *qh = q + hoff, *kh = k + hoff, *vh = v + hoff;
    const float *mrow = mask ? mask + (long)b * T : nullptr;
    const float scale = sqrtf((float)dk);
    ATT_STAMP(0);

    // ---- 0. small operands into LDS: key mask, relative tables (every later phase reads them from LDS: the per-element
    //         global loads of the first version of this kernel cost a full memory latency each) -------------------------
    for (int c = tid; c < ntiles * 32; c += kAttThreads) Ms[c] = (mrow && c < T) ? mrow[c] : 1.f;
    for (int e = tid; e < nrel * DK; e += kAttThreads) {
        const int r = e / DK, c = e - r * DK;
        EkL[e] = (c < dk) ? emb_k[r * dk + c] : 0.f;
        EvL[e] = (c < dk) ? emb_v[r * dk + c] : 0.f;
    }

    // ---- 1. S = Q K^T / sqrt(dk) ------------------------------------------------------------
    // Q fragment of this block's 32 query rows: lane (j, hh) holds Q[channel 2ks+hh][t0+j].  All loads of a tile are
    // issued before its MFMAs (addresses clamped, invalid lanes zeroed by a select: no branch per element).
    // Buffer loads: an invalid lane gets the out-of-range offset and reads 0 from the hardware range check — no branch,
    // no select on the data (a plain `cond ? load : 0` makes hipcc emit a branch + full wait per element).
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
    for (int jt = wave; jt < ntiles; jt += 4) {
        const int col = jt * 32 + j;
        const bool kv = col < T;
        float bk[DK / 2];
#pragma unroll
        for (int ks = 0; ks < DK / 2; ++ks) {
            const int ch = 2 * ks + hh;
            bk[ks] = ld_buf(rk, (kv && ch < dk) ? (ch * T + col) * 4 : kBufOob, 0);
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < DK / 2; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[ks], bk[ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
            S[row * pitch + col] = acc[r] / scale;
        }
    }
    __syncthreads();
    ATT_STAMP(1);

    // ---- 2a. relative-key band: S[i][i+d] += (Q[i] . Ek[d+w]) / sqrt(dk),  |d| <= w ----------
    // from the Q fragment already in registers: each half-wave sums its 48 channels, one shuffle joins the halves
    for (int r = wave; r < nrel; r += 4) {
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < DK / 2; ++ks) part += aq[ks] * EkL[r * DK + 2 * ks + hh];
        const float dot = part + __shfl_xor(part, 32);
        const int ti = t0 + j;
        const int tj = ti + r - window;
        if (hh == 0 && ti < T && tj >= 0 && tj < T) S[j * pitch + tj] += dot / scale;
    }
    if (nrel) __syncthreads();

    ATT_STAMP(2);
    // ---- 2b. mask fill + softmax: each wave owns 8 rows and walks them together (8 independent LDS chains per
    //          column step instead of one row at a time).  Rows beyond T hold S = 0 (their Q fragment is zero): they
    //          normalise to 1/T, stay finite and are never stored.
    {
        float *Sw = S + wave * 8 * pitch;
        float mi[8], mx[8], sum[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            mi[rr] = Ms[t0 + wave * 8 + rr];
            mx[rr] = -INFINITY;
            sum[rr] = 0.f;
        }
        for (int c = lane; c < T; c += 64) {
            const float mc = Ms[c];
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                float sv = Sw[rr * pitch + c];
                if (mrow && (mi[rr] == 0.f || mc == 0.f)) sv = -1e4f;
                Sw[rr * pitch + c] = sv;
                mx[rr] = fmaxf(mx[rr], sv);
            }
        }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) mx[rr] = wave_max(mx[rr]);
        for (int c = lane; c < T; c += 64) {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const float ev = expf(Sw[rr * pitch + c] - mx[rr]);
                Sw[rr * pitch + c] = ev;
                sum[rr] += ev;
            }
        }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) sum[rr] = wave_sum(sum[rr]);
        for (int c = lane; c < ntiles * 32; c += 64) {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) Sw[rr * pitch + c] = (c < T) ? Sw[rr * pitch + c] / sum[rr] : 0.f;
        }
    }
    __syncthreads();
    ATT_STAMP(3);

    // ---- 3. O^T[n][i] = sum_kk V^T[n][kk] P^T[kk][i] ------------------------------------------
    // The pairing of contracted indices with (k-step, half-wave) is free as long as both operands use the same one:
    // with kk = 16*hh + ks a lane's 16 V values of a key tile are CONSECUTIVE in memory (row n, columns kt*32+16hh..+15),
    // so V^T fragments come straight from global memory (next tile prefetched) — no LDS staging, no barriers in this loop.
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool mma_wave = wave < DK / 32;
    if (mma_wave) {
        const int n = wave * 32 + j;                 // rows >= dk read 0 and are never stored
        float vv[2][16];
        auto vload = [&](int kt, float (&dst)[16]) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int col = kt * 32 + 16 * hh + m;
                dst[m] = ld_buf(rv, (n < dk && col < T) ? (n * T + col) * 4 : kBufOob, 0);
            }
        };
        vload(0, vv[0]);
        for (int kt = 0; kt < ntiles; kt += 2) {
            if (kt + 1 < ntiles) vload(kt + 1, vv[1]);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[0][ks], S[j * pitch + kt * 32 + 16 * hh + ks], acc, 0, 0, 0);
            if (kt + 1 < ntiles) {
                if (kt + 2 < ntiles) vload(kt + 2, vv[0]);
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[1][ks], S[j * pitch + (kt + 1) * 32 + 16 * hh + ks], acc, 0, 0, 0);
            }
        }
    }

    ATT_STAMP(4);
    // ---- 4. relative-value band + store ---------------------------------------------------------
    if (mma_wave) {
        const int ti = t0 + j;
        if (ti < T) {
            float rel[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) rel[r] = 0.f;
            for (int d = 0; d < nrel; ++d) {
                const int tj = ti + d - window;
                const float p = (tj >= 0 && tj < T) ? S[j * pitch + tj] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) rel[r] += p * EvL[d * DK + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (n < dk) out[((long)b * heads * dk + (long)head * dk + n) * T + ti] = acc[r] + rel[r];
            }
        }
    }
    ATT_STAMP(5);
}


// ---- any T (T > 1024: the [32][T] score strip no longer fits in LDS) ---------------------------------------------------
// transformer.py:118-163 has no bound on T.  Same block = 32 query rows of one (batch, head), same fp32 MFMA contractions,
// but the scores are never held for the whole row: pass 1 walks the key tiles (4 per iteration, one per wave) and keeps
// only the running row maximum and sum (online softmax), pass 2 recomputes each score tile, turns it into probabilities
// p = exp(s - max) / sum in four [32][33] LDS tiles, and the channel waves consume those for O^T += V^T P^T and the
// relative-value band.  QK^T is computed twice — a fallback for long sequences, not the fast path.  Differs from the strip
// kernel only in the summation order of the softmax denominator.
template <int DK>
__global__ __launch_bounds__(kAttThreads, 2) void rel_attention_long_kernel(
    float *__restrict__ out, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, long qkv_bstride, const float *__restrict__ mask,
    const float *__restrict__ emb_k, const float *__restrict__ emb_v, int window, int heads, int dk, int T)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int kP = 33;
    const int ntiles = (T + 31) / 32;
    const int niter = (ntiles + 3) / 4;
    const int nrel = emb_k ? 2 * window + 1 : 0;
    float *Pt = smem;                          // [4 waves][32][33] score / probability tiles
    float *St = Pt + 4 * 32 * kP;              // [2][4][32] per-wave row maxima / sums, then [2][32] merged
    float *EkL = St + 2 * 4 * 32;              // [nrel][DK]
    float *EvL = EkL + nrel * DK;              // [nrel][DK]
    float *Dk = EvL + nrel * DK;               // [nrel][32] relative-key logits of this block's rows (already / sqrt(dk))
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5;
    const int j = lane & 31;
    const int t0 = blockIdx.x * kAttRows;
    const int head = blockIdx.y;
    const int b = blockIdx.z;
    const long hoff = (long)b * qkv_bstride + (long)head * dk * T;
    const float *qh = q + hoff, *kh = k + hoff, *vh = v + hoff;
    const float *mrow = mask ? mask + (long)b * T : nullptr;
    const float scale = sqrtf((float)dk);
    float *Pw = Pt + wave * 32 * kP;

    for (int e = tid; e < nrel * DK; e += kAttThreads) {
        const int r = e / DK, c = e - r * DK;
        EkL[e] = (c < dk) ? emb_k[r * dk + c] : 0.f;
        EvL[e] = (c < dk) ? emb_v[r * dk + c] : 0.f;
    }
    const long slab = (long)dk * T * 4;
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(qh, slab), rk = make_rsrc(kh, slab), rv = make_rsrc(vh, slab);
    float aq[DK / 2];
    {
        const bool qv = (t0 + j) < T;
#pragma unroll
        for (int ks = 0; ks < DK / 2; ++ks) {
            const int ch = 2 * ks + hh;
            aq[ks] = ld_buf(rq, (qv && ch < dk) ? (int)(((long)ch * T + t0 + j) * 4) : kBufOob, 0);
        }
    }
    __syncthreads();
    for (int r = wave; r < nrel; r += 4) {
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < DK / 2; ++ks) part += aq[ks] * EkL[r * DK + 2 * ks + hh];
        const float dot = part + __shfl_xor(part, 32);
        if (hh == 0) Dk[r * 32 + j] = dot / scale;
    }
    const float mi = (mrow && t0 + j < T) ? mrow[t0 + j] : 1.f;     // query-row mask of row j (both half-waves)
    __syncthreads();

    // masked scores of key tile kt into this wave's LDS tile (rows = queries, columns = keys of the tile)
    auto score_tile = [&](int kt) {
        const int col = kt * 32 + j;
        const bool kv = (kt < ntiles) && col < T;
        float bk[DK / 2];
#pragma unroll
        for (int ks = 0; ks < DK / 2; ++ks) {
            const int ch = 2 * ks + hh;
            bk[ks] = ld_buf(rk, (kv && ch < dk) ? (int)(((long)ch * T + col) * 4) : kBufOob, 0);
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < DK / 2; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[ks], bk[ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) Pw[((r & 3) + 8 * (r >> 2) + 4 * hh) * kP + j] = acc[r] / scale;
    };
    // band + mask fill on the wave's tile; lane (j, hh): row j, columns 16hh..16hh+15.  Returns nothing; tile stays in LDS.
    auto band_mask = [&](int kt) {
        for (int r = hh; r < nrel; r += 2) {
            const int tj = t0 + j + r - window;
            if (t0 + j < T && tj >= kt * 32 && tj < kt * 32 + 32 && tj < T) Pw[j * kP + tj - kt * 32] += Dk[r * 32 + j];
        }
    };

    // ---- pass 1: running row maximum / sum ------------------------------------------------------------------------------
    float m_run = -INFINITY, l_run = 0.f;
    for (int it = 0; it < niter; ++it) {
        const int kt = it * 4 + wave;
        score_tile(kt);
        __syncthreads();
        band_mask(kt);
        __syncthreads();
        if (kt < ntiles) {
            float sv[16];
            float tmax = -INFINITY;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int col = kt * 32 + 16 * hh + c;
                float x = Pw[j * kP + 16 * hh + c];
                if (mrow && (mi == 0.f || mrow[col < T ? col : 0] == 0.f)) x = -1e4f;
                sv[c] = (col < T) ? x : -INFINITY;
                tmax = fmaxf(tmax, sv[c]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float m_new = fmaxf(m_run, tmax);         // finite: column kt*32 < T is valid
            float ps = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) ps += expf(sv[c] - m_new);
            ps += __shfl_xor(ps, 32);
            l_run = l_run * expf(m_run - m_new) + ps;       // first tile: 0 * exp(-inf) = 0
            m_run = m_new;
        }
        __syncthreads();
    }
    if (hh == 0) {
        St[wave * 32 + j] = m_run;
        St[4 * 32 + wave * 32 + j] = l_run;
    }
    __syncthreads();
    float m_tot = St[j];
#pragma unroll
    for (int w = 1; w < 4; ++w) m_tot = fmaxf(m_tot, St[w * 32 + j]);
    float l_tot = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float mw = St[w * 32 + j];
        if (mw != -INFINITY) l_tot += St[4 * 32 + w * 32 + j] * expf(mw - m_tot);
    }
    __syncthreads();

    // ---- pass 2: probabilities tile by tile, O^T += V^T P^T, relative-value band ----------------------------------------
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float rel[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rel[r] = 0.f;
    const bool mma_wave = wave < DK / 32;
    const int n = wave * 32 + j;
    const int ti = t0 + j;
    for (int it = 0; it < niter; ++it) {
        const int kt = it * 4 + wave;
        score_tile(kt);
        __syncthreads();
        band_mask(kt);
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int col = kt * 32 + 16 * hh + c;
            float x = Pw[j * kP + 16 * hh + c];
            if (mrow && (mi == 0.f || mrow[col < T ? col : 0] == 0.f)) x = -1e4f;
            Pw[j * kP + 16 * hh + c] = (kt < ntiles && col < T) ? expf(x - m_tot) / l_tot : 0.f;
        }
        __syncthreads();
        if (mma_wave) {
            for (int w = 0; w < 4; ++w) {
                const int kw = it * 4 + w;
                if (kw >= ntiles) break;
                const float *P = Pt + w * 32 * kP;
                float vv[16];
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    const int col = kw * 32 + 16 * hh + m;
                    vv[m] = ld_buf(rv, (n < dk && col < T) ? (int)(((long)n * T + col) * 4) : kBufOob, 0);
                }
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[ks], P[j * kP + 16 * hh + ks], acc, 0, 0, 0);
                for (int d = 0; d < nrel; ++d) {
                    const int tj = ti + d - window;
                    if (tj >= kw * 32 && tj < kw * 32 + 32 && tj < T) {
                        const float p = P[j * kP + tj - kw * 32];
#pragma unroll
                        for (int r = 0; r < 16; ++r) rel[r] += p * EvL[d * DK + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh];
                    }
                }
            }
        }
        __syncthreads();
    }
    if (mma_wave && ti < T) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nn = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (nn < dk) out[((long)b * heads * dk + (long)head * dk + nn) * T + ti] = acc[r] + rel[r];
        }
    }
}

}  // namespace ttsamd
#include "attention_v2.h"
#include "attention_v3.h"
namespace ttsamd {

#ifndef TTSAMD_ATT_V3_DEFAULT
#define TTSAMD_ATT_V3_DEFAULT true
#endif
constexpr long kAttV2MaxBlocks = 96;     // (query tile, head, item) blocks up to which the 8-wave small-grid kernel is taken

template <int DK>
static int launch_att(float *out, const float *q, const float *k, const float *v, long bstride, const float *mask,
                      const float *ek, const float *ev, int window, int batch, int heads, int dk, int T, hipStream_t st)
{
    const int ntiles = (T + 31) / 32;
    const int pitch = ntiles * 32 + 1;
    const int nrel = ek ? 2 * window + 1 : 0;
    const size_t lds = (size_t)(kAttRows * pitch + ntiles * 32 + 2 * nrel * DK) * sizeof(float);
    static const bool force_long = getenv("TTSAMD_ATT_FORCE_LONG") != nullptr;     // test hook (tests/test_text_gpu.py)
    if (lds > 160 * 1024 || T > 1024 || force_long) {
        // the score strip does not fit: tile-by-tile online-softmax kernel (any T)
        const size_t lds2 = (size_t)(4 * 32 * 33 + 2 * 4 * 32 + 2 * nrel * DK + nrel * 32) * sizeof(float);
        if (lds2 > 160 * 1024) {
            set_error("rel_attention: window=%d needs %zu bytes of LDS", window, lds2);
            return TTSAMD_ERR_UNSUPPORTED;
        }
        auto kl = rel_attention_long_kernel<DK>;
        static std::atomic<unsigned long long> lds_attr_long{0};
        TTSAMD_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(kl), (int)(160 * 1024), lds_attr_long));
        hipLaunchKernelGGL(kl, dim3(ntiles, heads, batch), dim3(kAttThreads), lds2, st, out, q, k, v, bstride, mask, ek, ev,
                           window, heads, dk, T);
        TTSAMD_LAUNCH_CHECK();
        return TTSAMD_OK;
    }
    {
        // 16-query blocks (attention_v3.h): TTSAMD_ATT_V3=0 / 1 forces the choice (tests run every kernel on the same inputs)
        static const char *force_v3 = getenv("TTSAMD_ATT_V3");
        const size_t lds_v3 = att3::lds_bytes(T, nrel, DK);
        if ((force_v3 ? force_v3[0] == '1' : TTSAMD_ATT_V3_DEFAULT) && lds_v3 <= 160 * 1024) {
            auto k3 = att3::rel_attention_v3_kernel<DK>;
            static std::atomic<unsigned long long> lds_attr_v3{0};
            TTSAMD_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(k3), (int)(160 * 1024), lds_attr_v3));
            hipLaunchKernelGGL(k3, dim3((T + 15) / 16, heads, batch), dim3(att3::kThreads), lds_v3, st, out, q, k, v, bstride, mask,
                               ek, ev, window, heads, dk, T);
            TTSAMD_LAUNCH_CHECK();
            return TTSAMD_OK;
        }
    }
    {
        // small grids (a single request: 18 blocks at T = 257): the 8-wave kernel of attention_v2.h.  TTSAMD_ATT_V2=0 / 1 forces
        // the choice (tests run both kernels on the same inputs)
        static const char *force_v2 = getenv("TTSAMD_ATT_V2");
        // (from five key tiles up: at T = 64 — two key tiles, six of its eight waves without one — it measured 36 us against the
        // 4-wave kernel's 16)
        const bool small = (long)ntiles * heads * batch <= kAttV2MaxBlocks && ntiles >= 5;
        const size_t lds_v2 = (size_t)(att2::kRows * (ntiles * 32 + 4) + ntiles * 32 + 2 * nrel * DK + att2::kWaves * 1024) * sizeof(float);
        if ((force_v2 ? force_v2[0] == '1' : small) && lds_v2 <= 160 * 1024) {
            auto k2 = att2::rel_attention_v2_kernel<DK>;
            static std::atomic<unsigned long long> lds_attr_v2{0};
            TTSAMD_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(k2), (int)(160 * 1024), lds_attr_v2));
            hipLaunchKernelGGL(k2, dim3(ntiles, heads, batch), dim3(att2::kThreads), lds_v2, st, out, q, k, v, bstride, mask, ek, ev,
                               window, heads, dk, T, ntiles * 32 + 4);
            TTSAMD_LAUNCH_CHECK();
            return TTSAMD_OK;
        }
    }
    auto kern = rel_attention_kernel<DK>;
    static std::atomic<unsigned long long> lds_attr_done{0};   // per device, see ensure_dynamic_lds
    TTSAMD_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), (int)(160 * 1024), lds_attr_done));
    hipLaunchKernelGGL(kern, dim3(ntiles, heads, batch), dim3(kAttThreads), lds, st, out, q, k, v, bstride, mask,
                       ek, ev, window, heads, dk, T, pitch);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

}  // namespace ttsamd
using namespace ttsamd;

#ifdef TTSAMD_PHASE_CLOCKS
extern "C" int ttsamd_debug_att_clocks(long long *host_out8)
{
    return hipMemcpyFromSymbol(host_out8, HIP_SYMBOL(ttsamd::g_att_clk), 8 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int ttsamd_rel_attention(float *out, const float *q, const float *k, const float *v, int64_t qkv_bstride,
                                    const float *mask, const float *emb_rel_k, const float *emb_rel_v, int window,
                                    int batch, int heads, int dk, int t, void *stream)
{
    TTSAMD_CHECK_ARG(out && q && k && v, "rel_attention: NULL tensor");
    TTSAMD_CHECK_ARG(batch >= 0 && heads > 0 && dk > 0 && t >= 0, "rel_attention: bad shape");
    TTSAMD_CHECK_ARG((emb_rel_k == nullptr) == (emb_rel_v == nullptr), "rel_attention: need both or neither rel embeddings");
    TTSAMD_CHECK_ARG(!emb_rel_k || window >= 0, "rel_attention: bad window");
    if (batch == 0 || t == 0) return TTSAMD_OK;
    if (dk > 128 || batch > 65535 || heads > 65535 || (int64_t)dk * t * 4 >= 0x7FFFFFF0ll) {
        set_error("rel_attention: unsupported shape (dk=%d <= 128, one head's [dk, T] slab < 2 GiB)", dk);
        return TTSAMD_ERR_UNSUPPORTED;
    }
    hipStream_t st = as_stream(stream);
    switch ((dk + 31) / 32) {   // e.g. dk = 98: multilingual VITS, (192 + 4 language channels) / 2 heads
        case 1: return launch_att<32>(out, q, k, v, qkv_bstride, mask, emb_rel_k, emb_rel_v, window, batch, heads, dk, t, st);
        case 2: return launch_att<64>(out, q, k, v, qkv_bstride, mask, emb_rel_k, emb_rel_v, window, batch, heads, dk, t, st);
        case 3: return launch_att<96>(out, q, k, v, qkv_bstride, mask, emb_rel_k, emb_rel_v, window, batch, heads, dk, t, st);
        default: return launch_att<128>(out, q, k, v, qkv_bstride, mask, emb_rel_k, emb_rel_v, window, batch, heads, dk, t, st);
    }
}
