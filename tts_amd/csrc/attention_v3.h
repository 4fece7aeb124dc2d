// Relative-position attention on 16-query blocks (round 6).  Same operator and arithmetic as rel_attention_kernel /
// rel_attention_v2_kernel (transformer.py:118-163 and the skew helpers :196-241; exact fp32 MFMA products, scores divided by
// sqrt(dk) after the contraction, -1e4 mask fill, expf) — only summation orders change.
//
// Why another cut: a 32-query block of one (item, head) is 2 x 1.77 MFLOP of exact-fp32 MFMA work = 14 k cycles on ONE CU's
// four fp32 matrix pipes before any latency is paid, and a single request has 18 such blocks on a 256-CU chip (28-31 us per
// launch, six launches per VITS request, profiles/r06_b1_timeline.txt).  Here
//   * a block is 16 query rows (v_mfma_f32_16x16x4_f32): twice the blocks, half the serial matrix work per block;
//   * Q K^T: 16-column tiles dealt round-robin to the waves, a wave's first tiles requested before anything else; every load
//     is straight-line code: the lane's column offset in the vector operand, the channel step in the scalar operand, absent
//     columns / channels out of the buffer's range (`cond ? load(a) : load(b)` per element makes hipcc emit two divergent paths
//     with a full wait between consecutive loads: the 32-query kernels' contraction was a chain of memory round trips);
//   * the relative-key logits R = Q Ek^T are ONE more 16x16 MFMA tile dealt out with the key tiles, kept in LDS and added by
//     the wave that owns the row just before its softmax: no scatter pass over the block, one barrier less;
//   * softmax: a wave owns 2 rows and keeps them in registers between its passes; reductions on the DPP network;
//   * P V: wave = (half of the channel tiles, every 4th 16-key step): one ds_read_b128 of P feeds DK / 32 x 4 MFMAs, V
//     fragments are 16-byte buffer loads requested BEFORE the softmax; the four partial sums of an output meet in LDS in a
//     fixed order and ALL 512 threads run the relative-value band and the store.
// T <= 1024 (the score strip lives in LDS); longer sequences keep rel_attention_long_kernel.  Included by attention.hip only.
#pragma once

#ifndef TTSAMD_ATT3_KRING
#define TTSAMD_ATT3_KRING 2
#endif
#ifndef TTSAMD_ATT3_VRING
#define TTSAMD_ATT3_VRING 2
#endif
#ifndef TTSAMD_ATT3_VEARLY
#define TTSAMD_ATT3_VEARLY 1
#endif
namespace att3 {
using namespace ttsamd;

constexpr int kRows = 16;
constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr int kMaxSteps = 16;   // 64-column steps of a row: T <= 1024
constexpr int kKG = 4;          // key groups of P.V (wave = kg * 2 + channel half)
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x4u = __attribute__((ext_vector_type(4), aligned(4))) float;


// wave-wide reductions on the DPP network (four row steps, then the four rows through readlane): wave-uniform result.
// (attention.hip's wave_max / wave_sum are six dependent ds_bpermute round trips each)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_wave_max(float v)
{
    v = fmaxf(v, dpp_f32<0xB1>(v));     // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_f32<0x4E>(v));     // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_f32<0x141>(v));    // row_half_mirror
    v = fmaxf(v, dpp_f32<0x140>(v));    // row_mirror: every lane = its row's maximum
    return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)),
                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16))),
                 fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)),
                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48))));
}
__device__ __forceinline__ float dpp_wave_sum(float v)
{
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48)));
}

__host__ __device__ inline int rel_pitch(int nrel) { return ((nrel + 15) / 16) * 16 + 1; }

inline size_t lds_bytes(int T, int nrel, int DK)
{
    const int tp = ((T + 15) / 16) * 16;
    return (size_t)(kRows * (tp + 4) + tp + nrel * DK + kRows * rel_pitch(nrel) + kKG * DK * kRows) * sizeof(float);
}

template <int DK>  // dk rounded up to a multiple of 32; channels dk..DK-1 are treated as zeros
__global__ __launch_bounds__(kThreads) void rel_attention_v3_kernel(
    float *__restrict__ out, const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    long qkv_bstride, const float *__restrict__ mask, const float *__restrict__ emb_k, const float *__restrict__ emb_v,
    int window, int heads, int dk, int T)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NTW = DK / 32;                // 16-channel tiles per wave in P.V
    constexpr int KS = DK / 4;                  // k-steps of the Q K^T contraction
    const int nkt = (T + 15) / 16;
    const int tp = nkt * 16;
    const int pitch = tp + 4;                   // % 4 == 0: ds_read_b128 of P fragments
    const int nrel = emb_k ? 2 * window + 1 : 0;
    const int rp = rel_pitch(nrel);
    float *S = smem;                            // [16][pitch] scores / probabilities
    float *Ms = S + kRows * pitch;              // [tp] key mask (1 where absent)
    float *EvL = Ms + tp;                       // [nrel][DK] relative-value table, zero padded to DK
    float *Rl = EvL + nrel * DK;                // [16][rp] relative-key logits (already / sqrt(dk))
    float *Op = Rl + kRows * rp;                // [kKG][DK][16] partial outputs
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;                    // k index inside an MFMA step
    const int i = lane & 15;                    // row of an A fragment / column of a B fragment
    const int t0 = blockIdx.x * kRows;
    const int head = blockIdx.y;
    const int b = blockIdx.z;
    const long hoff = (long)b * qkv_bstride + (long)head * dk * T;
    const float *qh = q + hoff, *kh = k + hoff, *vh = v + hoff;
    const float *mrow = mask ? mask + (long)b * T : nullptr;
    const float scale = sqrtf((float)dk);

    // Loads: a lane's offset inside a [dk][T] slab is (g T + column) * 4 — or 2^31 when its column does not exist — in the
    // VECTOR offset, the k-step's 4 ks T * 4 in the SCALAR offset (the range check of a raw buffer access covers the sum:
    // scripts/ubench/soffset_range.hip).  Channels >= dk land beyond the slab and read 0 with no instruction spent on them;
    // written as `cond ? load(a) : load(b)` per element hipcc splits the wave into two paths with a full wait between them.
    constexpr int kInvalid = (int)0x80000000u;
    const int slab = dk * T * 4;
    const int kstep = 16 * T;                   // bytes between k-steps: 4 channels
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(qh, slab), rk = make_rsrc(kh, slab), rv = make_rsrc(vh, slab);

    ATT_STAMP(0);
    // ---- 1. S = Q K^T / sqrt(dk) and R = Q Ek^T / sqrt(dk): 16-column tiles dealt round-robin to the waves ---------------
    // "virtual" tile jt: a key tile for jt < nkt, then the tiles of the relative-key table.  One straight-line load sequence
    // serves both (descriptor, offsets and destination are wave-uniform selects); a wave's first three tiles are requested before
    // anything else happens, so the block pays one memory latency here.
    const int nrt = (nrel + 15) / 16;
    const int nvt = nkt + nrt;
    {
        float aq[KS];
        {
            const int qoff = (t0 + i) < T ? (g * T + t0 + i) * 4 : kInvalid;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) aq[ks] = ld_buf(rq, qoff, ks * kstep);
        }
        constexpr int kRing = TTSAMD_ATT3_KRING;
        float bk[kRing][KS];
        auto kload = [&](int jt, float(&dst)[KS]) {
            const bool isk = jt < nkt;
            const int col = jt * 16 + i;
            const int r = (jt - nkt) * 16 + i;
            // Ek[r][4 ks + g]: a channel >= dk aliases into the next row of the table (or beyond it: 0) and meets a zero of Q
            const int off = isk ? (col < T ? (g * T + col) * 4 : kInvalid) : ((jt < nvt && r < nrel) ? (r * dk + g) * 4 : kInvalid);
            const int step = isk ? kstep : 16;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(isk ? kh : emb_k, isk ? (long)slab : (long)nrel * dk * 4);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) dst[ks] = ld_buf(rs, off, ks * step);
        };
        auto ktile = [&](const float(&frag)[KS], int jt) {
            const bool isk = jt < nkt;
            float *dst = isk ? S + jt * 16 : Rl + (jt - nkt) * 16;
            const int dpitch = isk ? pitch : rp;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[ks], frag[ks], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(4 * g + r) * dpitch + i] = acc[r] / scale;
        };
#pragma unroll
        for (int u = 0; u < kRing; ++u) kload(wave + u * kWaves, bk[u]);
        // ---- 0. small operands into LDS, behind the first requests of the contraction ------------------------------------
        for (int c = tid; c < tp; c += kThreads) Ms[c] = (mrow && c < T) ? mrow[c] : 1.f;
        for (int e = tid; e < nrel * DK; e += kThreads) {
            const int r = e / DK, c = e - r * DK;
            EvL[e] = (c < dk) ? emb_v[r * dk + c] : 0.f;
        }
        ATT_STAMP(1);
        for (int jt = wave; jt < nvt; jt += kRing * kWaves) {
#pragma unroll
            for (int u = 0; u < kRing; ++u) {
                if (u == 0 || jt + u * kWaves < nvt) {
                    ktile(bk[u], jt + u * kWaves);
                    kload(jt + (u + kRing) * kWaves, bk[u]);
                }
            }
        }
    }
    ATT_STAMP(2);
    __syncthreads();
    ATT_STAMP(3);

    // P.V operands of this wave: wave -> (channel half cg, 16-key steps s = kg mod 4).  Its first four steps of V (and the one
    // partial step at the end of a row, element by element) are requested HERE, before the softmax, and consumed after it.
    constexpr int kVRing = TTSAMD_ATT3_VRING;
    const int cg = wave & 1;
    const int kg = wave >> 1;
    const int nfull = T >> 4;
    int voff[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int n = (cg * NTW + t) * 16 + i;          // channel of this lane's V^T row (rows >= dk are zeros)
        voff[t] = n < dk ? (n * T + 4 * g) * 4 : kInvalid;
    }
    f32x4 vv[kVRing][NTW];
    auto vload = [&](int s, f32x4(&dst)[NTW]) {
        const int so = s < nfull ? 64 * s : 0x7FFFFFF0;      // a step past the whole ones: out of range for every lane
#pragma unroll
        for (int t = 0; t < NTW; ++t) dst[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, voff[t], so, 0));
    };
    const bool tail = nfull < nkt && (nfull & (kKG - 1)) == kg;     // this wave owns the partial step
    float vt[NTW][4];
    auto vfirst = [&]() {
#pragma unroll
        for (int u = 0; u < kVRing; ++u) vload(kg + u * kKG, vv[u]);
        if (tail) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int eo = (16 * nfull + 4 * g + e < T) ? (16 * nfull + e) * 4 : kInvalid;
                // (arithmetic, not a select per load; 2^31 + 2^31 would wrap to a valid offset: the sign bit is or-ed back in)
#pragma unroll
                for (int t = 0; t < NTW; ++t) vt[t][e] = ld_buf(rv, (voff[t] + (eo & 0x7FFFFFFF)) | (eo & kInvalid), 0);
            }
        }
    };
    if constexpr (TTSAMD_ATT3_VEARLY) vfirst();

    // ---- 2. relative-key band + mask fill + softmax: a wave owns 2 rows and keeps them in registers between the passes ----
    {
        constexpr int RW = kRows / kWaves;   // 2
        float *Sw = S + wave * RW * pitch;
        const int nsteps = (T + 63) / 64;
        float mi[RW], mx[RW], sum[RW];
        float sv[RW][kMaxSteps];
        // S[i][i + d] += R[i][d + w], |d| <= w: the same wave reads these scores next (LDS executes a wave's accesses in order)
        for (int d = lane; d < nrel; d += 64) {
#pragma unroll
            for (int rr = 0; rr < RW; ++rr) {
                const int row = wave * RW + rr;
                const int tj = t0 + row + d - window;
                if (tj >= 0 && tj < T) Sw[rr * pitch + tj] += Rl[row * rp + d];
            }
        }
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
                const int cc = cv ? c : 0;
                const float mc = Ms[cc];
#pragma unroll
                for (int rr = 0; rr < RW; ++rr) {
                    float x = Sw[rr * pitch + cc];
                    if (mrow && (mi[rr] == 0.f || mc == 0.f)) x = -1e4f;
                    x = cv ? x : -INFINITY;
                    sv[rr][s] = x;
                    mx[rr] = fmaxf(mx[rr], x);
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) mx[rr] = dpp_wave_max(mx[rr]);
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
        for (int rr = 0; rr < RW; ++rr) sum[rr] = dpp_wave_sum(sum[rr]);
#pragma unroll
        for (int s = 0; s < kMaxSteps; ++s) {
            if (s < nsteps) {
                const int c = lane + 64 * s;
                if (c < tp) {
#pragma unroll
                    for (int rr = 0; rr < RW; ++rr) Sw[rr * pitch + c] = (c < T) ? sv[rr][s] / sum[rr] : 0.f;
                }
            }
        }
    }
    __syncthreads();
    ATT_STAMP(4);

    // ---- 3. O^T[n][i] = sum_kk V^T[n][kk] P^T[kk][i] ------------------------------------------------------------------------
    // The pairing of the contracted index with (k-step e, quarter-wave g) is free as long as both operands use the same one:
    // key = 16 s + 4 g + e makes a lane's four V values of a step consecutive in memory (one 16-byte buffer load, the step's
    // offset in the scalar operand) and its four P values one ds_read_b128.
    {
        f32x4 acc[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (!TTSAMD_ATT3_VEARLY) vfirst();
        auto ptile = [&](int s, const f32x4(&frag)[NTW]) {
            const f32x4 p4 = *reinterpret_cast<const f32x4 *>(S + i * pitch + 16 * s + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[t][e], p4[e], acc[t], 0, 0, 0);
        };
        for (int s = kg; s < nfull; s += kVRing * kKG) {
#pragma unroll
            for (int u = 0; u < kVRing; ++u) {
                if (u == 0 || s + u * kKG < nfull) {
                    ptile(s + u * kKG, vv[u]);
                    vload(s + (u + kVRing) * kKG, vv[u]);
                }
            }
        }
        if (tail) {
            const f32x4 p4 = *reinterpret_cast<const f32x4 *>(S + i * pitch + 16 * nfull + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(vt[t][e], p4[e], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) Op[(kg * DK + (cg * NTW + t) * 16 + 4 * g + r) * kRows + i] = acc[t][r];
    }
    ATT_STAMP(5);
    __syncthreads();
    ATT_STAMP(6);

    // ---- 4. partial sums (fixed order) + relative-value band + store: every thread, DK * 16 / 512 outputs of ONE query -------
    {
        const int qi = tid & (kRows - 1);
        const int ti = t0 + qi;
        constexpr int kMaxRelRegs = 16;          // probabilities of the band kept in registers up to this many diagonals
        float pb[kMaxRelRegs];
#pragma unroll
        for (int d = 0; d < kMaxRelRegs; ++d) {
            const int tj = ti + d - window;
            const bool ok = d < nrel && tj >= 0 && tj < T;
            const float p = S[qi * pitch + (ok ? tj : 0)];
            pb[d] = ok ? p : 0.f;
        }
#pragma unroll
        for (int e = tid; e < DK * kRows; e += kThreads) {
            const int n = e >> 4;
            float o = Op[e];
#pragma unroll
            for (int p = 1; p < kKG; ++p) o += Op[p * DK * kRows + e];
            float rel = 0.f;
            if (nrel <= kMaxRelRegs) {
#pragma unroll
                for (int d = 0; d < kMaxRelRegs; ++d)
                    if (d < nrel) rel += pb[d] * EvL[d * DK + n];
            } else {
                for (int d = 0; d < nrel; ++d) {
                    const int tj = ti + d - window;
                    const float p = (tj >= 0 && tj < T) ? S[qi * pitch + tj] : 0.f;
                    rel += p * EvL[d * DK + n];
                }
            }
            if (n < dk && ti < T) out[((long)b * heads * dk + (long)head * dk + n) * T + ti] = o + rel;
        }
    }
    ATT_STAMP(7);
}

}  // namespace att3
