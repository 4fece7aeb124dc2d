// Study for the next attention kernel (NOT part of the library build): the relative-position attention block of
// tts_amd/csrc/attention.hip re-cut for the phase times measured in round 2 (profiles/r02_attention_phase_clocks.txt,
// 89 k cycles per 32-query block at T = 257: QK^T 27 k, softmax 22.5 k, P.V 30 k, the rest 9 k):
//   * 8 waves per block instead of 4: the 9 key tiles of QK^T go one per wave (the next tile's K fragment is requested
//     before the current tile's 48 dependent fp32 MFMAs), P.V is split over (channel tile, key-tile subset) pairs and the
//     2-3 partial tiles of a channel tile meet in LDS in a fixed order;
//   * softmax keeps a wave's 4 rows in REGISTERS between its passes (one LDS read + one write per score instead of three
//     of each, and no read-after-write chains through LDS: the old loop's `Sw[..] = f(Sw[..])` serialises on aliasing);
//   * P fragments are ds_read_b128 (row pitch = 4 mod 32 floats: conflict free), V fragments 16-byte global loads.
// Same arithmetic as the library kernel (exact fp32 MFMA products, scores divided by sqrt(dk) after the contraction, -1e4
// mask fill, expf) — only summation orders inside P.V change.  Checked here against an fp64 CPU restatement and the
// library kernel; timing of both.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off att_v2.hip -o att_v2 && ./att_v2
#include "../../tts_amd/csrc/attention.hip"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace ttsamd {
void set_error(const char *, ...) {}
}

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

// ---- host: fp64 restatement, comparison, timing ---------------------------------------------------------------------
static void reference(std::vector<double> &o, const std::vector<float> &qkv, const std::vector<float> &mask, bool has_mask,
                      const std::vector<float> &ek, const std::vector<float> &ev, int window, int B, int H, int dk, int T)
{
    const int C = H * dk;
    o.assign((size_t)B * C * T, 0.0);
    std::vector<double> p(T);
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h) {
            const float *q = &qkv[((size_t)b * 3 * C + h * dk) * T];
            const float *k = q + (size_t)C * T, *v = q + (size_t)2 * C * T;
            for (int i = 0; i < T; ++i) {
                double mx = -1e300;
                for (int jn = 0; jn < T; ++jn) {
                    double s = 0;
                    for (int c = 0; c < dk; ++c) s += (double)q[c * T + i] * k[c * T + jn];
                    const int d = jn - i;
                    if (window >= 0 && !ek.empty() && d >= -window && d <= window)
                        for (int c = 0; c < dk; ++c) s += (double)q[c * T + i] * ek[(d + window) * dk + c];
                    s /= std::sqrt((double)dk);
                    if (has_mask && (mask[b * T + i] == 0.f || mask[b * T + jn] == 0.f)) s = -1e4;
                    p[jn] = s;
                    mx = std::max(mx, s);
                }
                double sum = 0;
                for (int jn = 0; jn < T; ++jn) sum += (p[jn] = std::exp(p[jn] - mx));
                for (int jn = 0; jn < T; ++jn) p[jn] /= sum;
                for (int c = 0; c < dk; ++c) {
                    double a = 0;
                    for (int jn = 0; jn < T; ++jn) a += p[jn] * v[c * T + jn];
                    if (!ev.empty())
                        for (int d = -window; d <= window; ++d)
                            if (i + d >= 0 && i + d < T) a += p[i + d] * ev[(d + window) * dk + c];
                    o[((size_t)b * C + h * dk + c) * T + i] = a;
                }
            }
        }
}

template <typename F>
static float time_us(F f, int n)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0, 0);
    for (int i = 0; i < n; ++i) f();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / n;
}

static int run_case(int B, int H, int dk, int T, int window, bool has_mask)
{
    const int C = H * dk;
    std::vector<float> qkv((size_t)B * 3 * C * T), mask((size_t)B * T, 1.f), ek, ev;
    srand(B * 131 + T * 7 + dk);
    auto rnd = [] { return (float)((rand() % 20001) - 10000) * 1e-4f; };
    for (auto &x : qkv) x = rnd();
    if (window >= 0) {
        ek.resize((size_t)(2 * window + 1) * dk);
        ev.resize(ek.size());
        for (auto &x : ek) x = rnd();
        for (auto &x : ev) x = rnd();
    }
    if (has_mask)
        for (int b = 0; b < B; ++b)
            for (int t = T - (b * 7) % (T / 2 + 1); t < T; ++t) mask[b * T + t] = 0.f;
    float *d_qkv, *d_mask, *d_ek = nullptr, *d_ev = nullptr, *d_o1, *d_o2;
    hipMalloc(&d_qkv, qkv.size() * 4);
    hipMalloc(&d_mask, mask.size() * 4);
    hipMalloc(&d_o1, (size_t)B * C * T * 4);
    hipMalloc(&d_o2, (size_t)B * C * T * 4);
    hipMemcpy(d_qkv, qkv.data(), qkv.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_mask, mask.data(), mask.size() * 4, hipMemcpyHostToDevice);
    if (window >= 0) {
        hipMalloc(&d_ek, ek.size() * 4);
        hipMalloc(&d_ev, ev.size() * 4);
        hipMemcpy(d_ek, ek.data(), ek.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(d_ev, ev.data(), ev.size() * 4, hipMemcpyHostToDevice);
    }
    const float *q = d_qkv, *k = d_qkv + (size_t)C * T, *v = d_qkv + (size_t)2 * C * T;
    const long bs = (long)3 * C * T;
    const float *mk = has_mask ? d_mask : nullptr;
    auto f1 = [&] { ttsamd_rel_attention(d_o1, q, k, v, bs, mk, d_ek, d_ev, window < 0 ? 0 : window, B, H, dk, T, nullptr); };
    auto f2 = [&] {
        switch ((dk + 31) / 32) {
            case 1: att2::launch<32>(d_o2, q, k, v, bs, mk, d_ek, d_ev, window, B, H, dk, T, nullptr); break;
            case 2: att2::launch<64>(d_o2, q, k, v, bs, mk, d_ek, d_ev, window, B, H, dk, T, nullptr); break;
            case 3: att2::launch<96>(d_o2, q, k, v, bs, mk, d_ek, d_ev, window, B, H, dk, T, nullptr); break;
            default: att2::launch<128>(d_o2, q, k, v, bs, mk, d_ek, d_ev, window, B, H, dk, T, nullptr); break;
        }
    };
    hipMemset(d_o1, 0xff, (size_t)B * C * T * 4);
    hipMemset(d_o2, 0xff, (size_t)B * C * T * 4);
    f1();
    f2();
    hipDeviceSynchronize();
    std::vector<float> o1((size_t)B * C * T), o2(o1.size());
    hipMemcpy(o1.data(), d_o1, o1.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(o2.data(), d_o2, o2.size() * 4, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0, nrm = 0;
    if ((size_t)B * H * T * T * dk < (size_t)4e9) {
        std::vector<double> ref;
        reference(ref, qkv, mask, has_mask, ek, ev, window, B, H, dk, T);
        for (size_t i = 0; i < ref.size(); ++i) {
            // rows of fully masked queries are compared too: both kernels and the restatement fill -1e4 there
            e1 = std::max(e1, std::abs(o1[i] - ref[i]));
            e2 = std::max(e2, std::abs(o2[i] - ref[i]));
            nrm = std::max(nrm, std::abs(ref[i]));
        }
    }
    double d12 = 0;
    for (size_t i = 0; i < o1.size(); ++i) d12 = std::max(d12, (double)std::abs(o1[i] - o2[i]));
    const float t1 = time_us(f1, 50), t2 = time_us(f2, 50);
    const bool ok = std::isfinite(e2) && e2 <= 2e-5 * std::max(1.0, nrm) && std::isfinite(d12);
    printf("B=%-3d H=%d dk=%-3d T=%-4d window=%-2d mask=%d | library %7.1f us  v2 %7.1f us | max err vs fp64: library %.2e  v2 %.2e  "
           "(|ref| max %.2f)  library-v2 %.2e  %s\n",
           B, H, dk, T, window, (int)has_mask, t1, t2, e1, e2, nrm, d12, ok ? "ok" : "MISMATCH");
    hipFree(d_qkv); hipFree(d_mask); hipFree(d_o1); hipFree(d_o2);
    if (d_ek) hipFree(d_ek);
    if (d_ev) hipFree(d_ev);
    return ok ? 0 : 1;
}

int main()
{
    int bad = 0;
    bad += run_case(1, 2, 96, 257, 4, true);
    bad += run_case(32, 2, 96, 257, 4, true);
    bad += run_case(1, 2, 96, 64, 4, false);
    bad += run_case(2, 2, 96, 3, 4, true);       // T < window + 1
    bad += run_case(3, 2, 98, 130, 4, true);     // dk = 98 -> DK = 128 (multilingual VITS)
    bad += run_case(2, 4, 48, 70, -1, true);     // no relative tables
    bad += run_case(1, 2, 32, 513, 10, false);
    bad += run_case(1, 2, 96, 1024, 4, true);
    printf(bad ? "FAILED: %d case(s)\n" : "all cases ok\n", bad);
    return bad;
}
