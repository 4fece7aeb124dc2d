// Whole-tile study of the split-bf16 conv (main loop + residual operand + output stores), synthetic operands:
//   hipcc --offload-arch=gfx950 -O3 x3_full.hip -o x3_full && ./x3_full
// MODE 0: one block per tile, residual folded into the accumulators' initial value (conv_kernel_x3.h today)
// MODE 1: one block per tile, residual loaded during the last tap group of the last chunk, added after the loop
// MODE 2: persistent blocks (grid = CUs x OCC) looping over tiles; during a tile's last tap group the block also
//         prefetches the NEXT tile's first weights and first activation chunk, so that the stores of tile i drain
//         under tile i+1's main loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
__device__ long g_clk[2];

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, long bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float ld_buf(__amdgpu_buffer_rsrc_t r, int voffset, int soffset)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voffset, soffset, 0));
}
__device__ __forceinline__ void st_buf(__amdgpu_buffer_rsrc_t r, float v, int voffset, int soffset)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voffset, soffset, 0);
}
constexpr int kOob = 0x7FFFFFF0;

__device__ __forceinline__ void split3(float x, unsigned &p1, unsigned &p2, unsigned &p3)
{
    const __bf16 a1 = (__bf16)x;
    const float r1 = x - (float)a1;
    const __bf16 a2 = (__bf16)r1;
    const float r2 = r1 - (float)a2;
    const __bf16 a3 = (__bf16)r2;
    p1 = __builtin_bit_cast(unsigned short, a1);
    p2 = __builtin_bit_cast(unsigned short, a2);
    p3 = __builtin_bit_cast(unsigned short, a3);
}

template <int K, int D, int MI, int NI, int WM, int WN, int OCC, int G, int MODE>
__global__ __launch_bounds__(64 * WM * WN, OCC) void x3_full(float *__restrict__ y, const float *__restrict__ res, const u32x4 *__restrict__ wpk,
                                                             const float *__restrict__ x, int c, int T, int nblk_n, int nblk_m, int ntiles)
{
    constexpr int kThreads = 64 * WM * WN;
    constexpr int kBN = 32 * NI * WN;
    constexpr int kXW = kBN + (K - 1) * D;
    constexpr int kPart = kXW * 32;
    constexpr int kBuf = 3 * kPart;
    constexpr int kItems = 2 * kXW;
    constexpr int kNSt = (kItems + kThreads - 1) / kThreads;
    constexpr int kNG = (K + G - 1) / G;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, h = lane >> 5, j = lane & 31;
    const int nchunks = c / 16;
    const long clk0 = clock64(), rt0 = wall_clock64();

    float st[kNSt][8];
    auto stage_load = [&](const float *xb, int t0, int chunk) {
#pragma unroll
        for (int i = 0; i < kNSt; ++i) {
            const int e = tid + i * kThreads;
            const int half = e / kXW, col = e - half * kXW;
            const int gt = t0 - (K - 1) * D / 2 + col;
            const bool ok = (e < kItems) && gt >= 0 && gt < T;
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(xb, (long)c * T * 4);
            const int vo = ok ? (half * 8 * T + gt) * 4 : kOob;
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) st[i][cc] = ld_buf(rx, vo, (chunk * 16 + cc) * T * 4);
        }
    };
    auto stage_store = [&](unsigned char *buf) {
#pragma unroll
        for (int i = 0; i < kNSt; ++i) {
            const int e = tid + i * kThreads;
            const int half = e / kXW, col = e - half * kXW;
            if (e < kItems) {
                unsigned p[3][8];
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                    float v = st[i][cc];
                    v = v > 0.f ? v : v * 0.1f;
                    split3(v, p[0][cc], p[1][cc], p[2][cc]);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    u32x4 w;
                    w.x = p[q][0] | (p[q][1] << 16); w.y = p[q][2] | (p[q][3] << 16);
                    w.z = p[q][4] | (p[q][5] << 16); w.w = p[q][6] | (p[q][7] << 16);
                    *reinterpret_cast<u32x4 *>(buf + q * kPart + col * 32 + half * 16) = w;
                }
            }
        }
    };
    auto decode = [&](int tile, int &b, int &mb, int &t0) {
        const int per_b = nblk_n * nblk_m;
        b = tile / per_b;
        const int r = tile - b * per_b;
        const int nb = r / nblk_m;
        mb = r - nb * nblk_m;
        t0 = nb * kBN;
    };

    int tile = blockIdx.x;
    const int tstride = (MODE == 2) ? gridDim.x : ntiles;   // classic grid: one tile per block
    int b, mb, t0;
    decode(tile, b, mb, t0);

    const u32x4 *wp[MI];
    auto set_wp = [&](int mblk) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const long mtile = ((long)mblk * WM + wm) * MI + mi;
            wp[mi] = wpk + mtile * ((long)nchunks * K * 3 * 64) + lane;
        }
    };
    set_wp(mb);
    u32x4 a_cur[G][MI][3], a_nxt[G][MI][3];
#pragma unroll
    for (int t = 0; t < G; ++t)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int q = 0; q < 3; ++q) a_cur[t][mi][q] = wp[mi][(long)t * 3 * 64 + q * 64];
    stage_load(x + (long)b * c * T, t0, 0);
    const int bbyte = (wn * (32 * NI) + j) * 32 + h * 16;
    int buf0 = 0;   // LDS buffer of chunk 0 of the current tile

    for (; tile < ntiles; tile += tstride) {
        f32x16 acc[MI][NI];
        float rr[MI][NI][16];
        const __amdgpu_buffer_rsrc_t rres = make_rsrc(res + (long)b * c * T, (long)c * T * 4);
        if (MODE == 0) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int row0 = ((mb * WM + wm) * MI + mi) * 32, t = t0 + wn * 32 * NI + ni * 32 + j;
                    const int vo = (t < T) ? (4 * h * T + t) * 4 : kOob;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = ld_buf(rres, vo, (row0 + (r & 3) + 8 * (r >> 2)) * T * 4);
                }
        } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        }
        stage_store(lds + buf0 * kBuf);
        __syncthreads();

        const int ntile = tile + tstride;
        int nb_ = b, nmb = mb, nt0 = t0;
        if (ntile < ntiles) decode(ntile, nb_, nmb, nt0);

        for (int cix = 0; cix < nchunks; ++cix) {
            const unsigned char *cur = lds + ((buf0 + cix) & 1) * kBuf;
            const bool last_chunk = (cix + 1 == nchunks);
#pragma unroll
            for (int g = 0; g < kNG; ++g) {
                const int tap0 = g * G;
                const int ntap = (K - tap0 < G) ? (K - tap0) : G;
                const int ntap0 = (g + 1 < kNG) ? (g + 1) * G : K;
                const int nn = (g + 1 < kNG) ? ((K - ntap0 < G) ? (K - ntap0) : G) : ((K < G) ? K : G);
                if (g == kNG - 1 && last_chunk) {
                    // next tile's first weights, its first activation chunk, and this tile's residual
                    if (ntile < ntiles) {
                        set_wp(nmb);
#pragma unroll
                        for (int t = 0; t < G; ++t)
                            if (t < nn)
#pragma unroll
                                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                                    for (int q = 0; q < 3; ++q) a_nxt[t][mi][q] = wp[mi][(long)t * 3 * 64 + q * 64];
                        stage_load(x + (long)nb_ * c * T, nt0, 0);
                    }
                    if (MODE != 0) {
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni) {
                                const int row0 = ((mb * WM + wm) * MI + mi) * 32, t = t0 + wn * 32 * NI + ni * 32 + j;
                                const int vo = (t < T) ? (4 * h * T + t) * 4 : kOob;
#pragma unroll
                                for (int r = 0; r < 16; ++r) rr[mi][ni][r] = ld_buf(rres, vo, (row0 + (r & 3) + 8 * (r >> 2)) * T * 4);
                            }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < G; ++t)
                        if (t < nn) {
                            const long gi = ((long)cix * K + ntap0 + t) * 3 * 64;
#pragma unroll
                            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                                for (int q = 0; q < 3; ++q) a_nxt[t][mi][q] = wp[mi][gi + q * 64];
                        }
                    if (g == kNG - 1) stage_load(x + (long)b * c * T, t0, cix + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < G; ++t)
                    if (t < ntap) {
                        const int tap = tap0 + t;
                        constexpr int pa[6] = {2, 1, 0, 1, 0, 0};
                        constexpr int pb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            u32x4 bq[3];
#pragma unroll
                            for (int q = 0; q < 3; ++q)
                                bq[q] = *reinterpret_cast<const u32x4 *>(cur + q * kPart + bbyte + (ni * 32 + tap * D) * 32);
#pragma unroll
                            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                                for (int u = 0; u < 6; ++u)
                                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_cur[t][mi][pa[u]]),
                                                                                          __builtin_bit_cast(bf16x8, bq[pb[u]]), acc[mi][ni], 0, 0, 0);
                        }
                    }
#pragma unroll
                for (int t = 0; t < G; ++t)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int q = 0; q < 3; ++q) a_cur[t][mi][q] = a_nxt[t][mi][q];
            }
            if (!last_chunk) stage_store(lds + ((buf0 + cix + 1) & 1) * kBuf);
            __syncthreads();
        }
        // epilogue: (+ residual) -> stores; they drain under the next tile's main loop in MODE 2
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(y + (long)b * c * T, (long)c * T * 4);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int row0 = ((mb * WM + wm) * MI + mi) * 32, t = t0 + wn * 32 * NI + ni * 32 + j;
                const int vo = (t < T) ? (4 * h * T + t) * 4 : kOob;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[mi][ni][r] + 0.01f;
                    if (MODE != 0) v += rr[mi][ni][r];
                    st_buf(ry, v, vo, (row0 + (r & 3) + 8 * (r >> 2)) * T * 4);
                }
            }
        buf0 = (buf0 + nchunks) & 1;
        b = nb_; mb = nmb; t0 = nt0;
    }
    if (tid == 0 && blockIdx.x == gridDim.x / 2) {
        g_clk[0] = clock64() - clk0;
        g_clk[1] = wall_clock64() - rt0;
    }
}

static float *g_y, *g_res, *g_x; static u32x4 *g_w;

template <int K, int D, int MI, int NI, int WM, int WN, int OCC, int G, int MODE>
void run(int c, int T, int batch = 32)
{
    constexpr int kBN = 32 * NI * WN, kBM = 32 * MI * WM;
    constexpr int kXW = kBN + (K - 1) * D;
    const size_t ldsb = (size_t)2 * 3 * kXW * 32;
    auto kern = x3_full<K, D, MI, NI, WM, WN, OCC, G, MODE>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern));
    const int nblk_n = (T + kBN - 1) / kBN, nblk_m = (c + kBM - 1) / kBM, ntiles = nblk_n * nblk_m * batch;
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(kern), 64 * WM * WN, ldsb);
    const int grid = (MODE == 2) ? (256 * occ < ntiles ? 256 * occ : ntiles) : ntiles;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), ldsb, 0, g_y, g_res, g_w, g_x, c, T, nblk_n, nblk_m, ntiles);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double flop = 2.0 * c * c * K * (double)T * batch;
    const double bytes = 3.0 * 4 * c * (double)T * batch;
    long hclk[2]; hipMemcpyFromSymbol(hclk, HIP_SYMBOL(g_clk), sizeof(hclk));
    hipError_t err = hipGetLastError();
    printf("[%.2f GHz] MODE%d <%d,%d,%d,%d> occ%d(%d) G=%d k=%2d C=%3d T=%6d grid=%6d vgpr=%3d spill=%d: %8.3f ms %7.1f TF-eq (%4.2f of bf16 peak) %6.2f TB/s %s\n",
           hclk[1] ? (double)hclk[0] / hclk[1] * 0.1 : 0.0, MODE, MI, NI, WM, WN, OCC, occ, G, K, c, T, grid, fa.numRegs, (int)fa.localSizeBytes, best,
           flop / best / 1e9, 6 * flop / best / 1e9 / 2500.0, bytes / best / 1e9, err == hipSuccess ? "" : hipGetErrorString(err));
    fflush(stdout);
}

template <int K, int G>
void suite()
{
    run<K, 1, 2, 2, 2, 2, 2, G, 0>(128, 49280);
    run<K, 1, 1, 4, 4, 1, 2, G, 0>(128, 49280);
    run<K, 1, 1, 4, 4, 1, 2, G, 1>(128, 49280);
    run<K, 1, 1, 4, 4, 1, 2, G, 2>(128, 49280);
    run<K, 1, 1, 4, 4, 1, 3, G, 0>(128, 49280);
    run<K, 1, 1, 4, 4, 1, 3, G, 2>(128, 49280);
    run<K, 1, 1, 4, 4, 1, 2, G, 0>(256, 6160);
    run<K, 1, 1, 4, 4, 1, 2, G, 2>(256, 6160);
    run<K, 1, 1, 4, 2, 2, 2, G, 0>(64, 98560);
    run<K, 1, 1, 4, 2, 2, 2, G, 1>(64, 98560);
    run<K, 1, 1, 4, 2, 2, 2, G, 2>(64, 98560);
    run<K, 1, 1, 2, 1, 4, 2, G, 0>(32, 197120);
    run<K, 1, 1, 2, 1, 4, 2, G, 1>(32, 197120);
    run<K, 1, 1, 2, 1, 4, 2, G, 2>(32, 197120);
    run<K, 1, 1, 2, 1, 4, 4, G, 2>(32, 197120);
}

int main(int argc, char **argv)
{
    const size_t wbytes = (size_t)64 << 20, xbytes = (size_t)1 << 30;
    hipMalloc(&g_w, wbytes); hipMalloc(&g_x, xbytes); hipMalloc(&g_y, xbytes); hipMalloc(&g_res, xbytes);
    std::vector<unsigned short> hw(wbytes / 2);
    for (auto &v : hw) v = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
    hipMemcpy(g_w, hw.data(), wbytes, hipMemcpyHostToDevice);
    std::vector<float> hx(xbytes / 4);
    for (auto &v : hx) v = (float)rand() / (float)RAND_MAX - 0.5f;
    hipMemcpy(g_x, hx.data(), xbytes, hipMemcpyHostToDevice);
    hipMemcpy(g_res, hx.data(), xbytes, hipMemcpyHostToDevice);
    suite<3, 3>();
    suite<11, 4>();
    suite<7, 4>();
    return 0;
}
