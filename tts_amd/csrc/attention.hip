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
//   3. O^T = V^T P^T               MFMA again; V chunks are staged through LDS (transposing the
//      contracted index onto the lane's k slot), P^T fragments come from the LDS strip.
//      Computing O^T (channels x time) makes the output stores coalesced along time.
//   4. band of relative-value terms added in registers, store.
#include "common.h"

namespace ttsamd {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kAttRows = 32;
constexpr int kAttThreads = 256;
constexpr int kAttVPitch = 33;

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
__global__ __launch_bounds__(kAttThreads) void rel_attention_kernel(
    float *__restrict__ out, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, long qkv_bstride, const float *__restrict__ mask,
    const float *__restrict__ emb_k, const float *__restrict__ emb_v, int window, int heads, int dk, int T, int pitch)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *S = smem;                            // [32][pitch]
    float *Vs = smem + kAttRows * pitch;        // [DK][kAttVPitch]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int hh = lane >> 5;
    const int j = lane & 31;
    const int t0 = blockIdx.x * kAttRows;
    const int head = blockIdx.y;
    const int b = blockIdx.z;
    const long hoff = (long)b * qkv_bstride + (long)head * dk * T;
    const float *qh = q + hoff, *kh = k + hoff, *vh = v + hoff;
    const float *mrow = mask ? mask + (long)b * T : nullptr;
    const float scale = sqrtf((float)dk);
    const int ntiles = (T + 31) / 32;

    // ---- 1. S = Q K^T / sqrt(dk) ------------------------------------------------------------
    {
        float aq[DK / 2];
        const bool qv = (t0 + j) < T;
#pragma unroll
        for (int ks = 0; ks < DK / 2; ++ks) aq[ks] = (qv && 2 * ks + hh < dk) ? qh[(long)(2 * ks + hh) * T + t0 + j] : 0.f;
        for (int jt = wave; jt < ntiles; jt += 4) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const int col = jt * 32 + j;
            const bool kv = col < T;
#pragma unroll
            for (int ks = 0; ks < DK / 2; ++ks) {
                const float bv = (kv && 2 * ks + hh < dk) ? kh[(long)(2 * ks + hh) * T + col] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[ks], bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
                S[row * pitch + col] = acc[r] / scale;
            }
        }
    }
    __syncthreads();

    // ---- 2a. relative-key band: S[i][i+d] += (Q[i] . Ek[d+w]) / sqrt(dk),  |d| <= w ----------
    if (emb_k) {
        const int nrel = 2 * window + 1;
        for (int idx = tid; idx < kAttRows * nrel; idx += kAttThreads) {
            const int i = idx & 31;
            const int r = idx >> 5;
            const int ti = t0 + i;
            const int tj = ti + r - window;
            if (ti < T && tj >= 0 && tj < T) {
                float dot = 0.f;
                for (int c = 0; c < dk; ++c) dot += qh[(long)c * T + ti] * emb_k[r * dk + c];
                S[i * pitch + tj] += dot / scale;
            }
        }
        __syncthreads();
    }

    // ---- 2b. mask fill + softmax (each wave owns 8 rows) --------------------------------------
    for (int rr = 0; rr < 8; ++rr) {
        const int i = wave * 8 + rr;
        const int ti = t0 + i;
        float *Srow = S + i * pitch;
        if (ti >= T) {  // query row outside the tensor: keep the strip finite, nothing is stored from it
            for (int c = lane; c < ntiles * 32; c += 64) Srow[c] = 0.f;
            continue;
        }
        const float mi = mrow ? mrow[ti] : 1.f;
        float mx = -INFINITY;
        for (int c = lane; c < T; c += 64) {
            float s = Srow[c];
            if (mrow && (mi == 0.f || mrow[c] == 0.f)) s = -1e4f;
            Srow[c] = s;
            mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int c = lane; c < T; c += 64) {
            const float e = expf(Srow[c] - mx);
            Srow[c] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        for (int c = lane; c < ntiles * 32; c += 64) Srow[c] = (c < T) ? Srow[c] / sum : 0.f;
    }
    __syncthreads();

    // ---- 3. O^T[n][i] = sum_kk V^T[n][kk] P^T[kk][i] ------------------------------------------
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool mma_wave = wave < DK / 32;
    for (int kt = 0; kt < ntiles; ++kt) {
        // stage V[:, kt*32 .. +32) -> Vs[n][c]   (coalesced along time, zero beyond T)
        for (int e = tid; e < DK * 32; e += kAttThreads) {
            const int n = e >> 5, c = e & 31;
            const int tt = kt * 32 + c;
            Vs[n * kAttVPitch + c] = (tt < T && n < dk) ? vh[(long)n * T + tt] : 0.f;
        }
        __syncthreads();
        if (mma_wave) {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const float av = Vs[(wave * 32 + j) * kAttVPitch + 2 * ks + hh];
                const float bv = S[j * pitch + kt * 32 + 2 * ks + hh];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- 4. relative-value band + store ---------------------------------------------------------
    if (mma_wave) {
        const int ti = t0 + j;
        if (ti < T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (n >= dk) continue;
                float o = acc[r];
                if (emb_v) {
                    float rel = 0.f;
                    for (int d = 0; d <= 2 * window; ++d) {
                        const int tj = ti + d - window;
                        if (tj >= 0 && tj < T) rel += S[j * pitch + tj] * emb_v[d * dk + n];
                    }
                    o += rel;
                }
                out[((long)b * heads * dk + (long)head * dk + n) * T + ti] = o;
            }
        }
    }
}

template <int DK>
static int launch_att(float *out, const float *q, const float *k, const float *v, long bstride, const float *mask,
                      const float *ek, const float *ev, int window, int batch, int heads, int dk, int T, hipStream_t st)
{
    const int ntiles = (T + 31) / 32;
    const int pitch = ntiles * 32 + 1;
    const size_t lds = (size_t)(kAttRows * pitch + DK * kAttVPitch) * sizeof(float);
    auto kern = rel_attention_kernel<DK>;
    static size_t lds_set = 0;
    if (lds > lds_set) {
        TTSAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        lds_set = 160 * 1024;
    }
    hipLaunchKernelGGL(kern, dim3(ntiles, heads, batch), dim3(kAttThreads), lds, st, out, q, k, v, bstride, mask,
                       ek, ev, window, heads, dk, T, pitch);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

}  // namespace ttsamd
using namespace ttsamd;

extern "C" int ttsamd_rel_attention(float *out, const float *q, const float *k, const float *v, int64_t qkv_bstride,
                                    const float *mask, const float *emb_rel_k, const float *emb_rel_v, int window,
                                    int batch, int heads, int dk, int t, void *stream)
{
    TTSAMD_CHECK_ARG(out && q && k && v, "rel_attention: NULL tensor");
    TTSAMD_CHECK_ARG(batch >= 0 && heads > 0 && dk > 0 && t >= 0, "rel_attention: bad shape");
    TTSAMD_CHECK_ARG((emb_rel_k == nullptr) == (emb_rel_v == nullptr), "rel_attention: need both or neither rel embeddings");
    TTSAMD_CHECK_ARG(!emb_rel_k || window >= 0, "rel_attention: bad window");
    if (batch == 0 || t == 0) return TTSAMD_OK;
    if (t > 1024 || dk > 128 || batch > 65535 || heads > 65535) {
        set_error("rel_attention: unsupported shape (dk=%d <= 128, T=%d <= 1024)", dk, t);
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
