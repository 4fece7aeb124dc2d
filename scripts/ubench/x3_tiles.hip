// Tile-shape / knock-out study of the split-bf16 conv main loop (conv_kernel_x3.h) on synthetic operands:
//   hipcc --offload-arch=gfx950 -O3 x3_tiles.hip -o x3_tiles && ./x3_tiles
// Geometry <MI,NI,WM,WN>: a wave owns MI x NI 32x32 tiles; A (weights) = MI*3 16-byte loads per lane per tap from L2,
// B (activations) = NI*3 ds_read_b128 per tap.  FLAGS: 1 = no A loads in the loop, 2 = no B LDS reads, 4 = no staging.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
__device__ long g_clk[2];

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

template <int K, int D, int MI, int NI, int WM, int WN, int OCC, int FLAGS>
__global__ __launch_bounds__(64 * WM * WN, OCC) void x3_loop(float *out, const u32x4 *__restrict__ wpk, const float *__restrict__ x,
                                                             int nchunks, int x_rstride, int t_in)
{
    constexpr bool NOA = FLAGS & 1, NOB = FLAGS & 2, STAGE = !(FLAGS & 4);
    constexpr int kThreads = 64 * WM * WN;
    constexpr int kBN = 32 * NI * WN;
    constexpr int kXW = kBN + (K - 1) * D;
    constexpr int kPart = kXW * 32;
    constexpr int kBuf = 3 * kPart;
    constexpr int kItems = 2 * kXW;
    constexpr int kNSt = (kItems + kThreads - 1) / kThreads;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, h = lane >> 5, j = lane & 31;
    const int t0 = blockIdx.x * kBN;
    const long clk0 = clock64(), rt0 = wall_clock64();

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const u32x4 *wp[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const long mtile = ((long)blockIdx.y * WM + wm) * MI + mi;
        wp[mi] = wpk + mtile * ((long)nchunks * K * 3 * 64) + lane;
    }
    float st[kNSt][8];
    auto stage_load = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < kNSt; ++i) {
            const int e = tid + i * kThreads;
            const int half = e / kXW, col = e - half * kXW;
            const int gt = t0 + col;
            const bool ok = (e < kItems) && gt < t_in;
            const float *src = x + ((long)(chunk * 16 + half * 8)) * x_rstride + gt;
#pragma unroll
            for (int c = 0; c < 8; ++c) st[i][c] = ok ? src[(long)c * x_rstride] : 0.f;
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
                for (int c = 0; c < 8; ++c) {
                    float v = st[i][c];
                    v = v > 0.f ? v : v * 0.1f;
                    split3(v, p[0][c], p[1][c], p[2][c]);
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
    if (STAGE) {
        stage_load(0);
        stage_store(lds);
    } else {
        for (int e = tid; e < 2 * kBuf / 4; e += kThreads) reinterpret_cast<unsigned *>(lds)[e] = 0x3c003c00u + e;
    }
    __syncthreads();

    const int bbyte = (wn * (32 * NI) + j) * 32 + h * 16;
    u32x4 a_cur[MI][3], a_nxt[MI][3];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int q = 0; q < 3; ++q) a_cur[mi][q] = wp[mi][q * 64];
    u32x4 bconst[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) bconst[q] = *reinterpret_cast<const u32x4 *>(lds + q * kPart + bbyte);

    for (int c = 0; c < nchunks; ++c) {
        const unsigned char *cur = lds + (STAGE ? (c & 1) * kBuf : 0);
        if (STAGE && c + 1 < nchunks) stage_load(c + 1);
#pragma unroll
        for (int tap = 0; tap < K; ++tap) {
            if (!NOA) {
                const long g = ((long)c * K + tap + 1) * 3 * 64;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int q = 0; q < 3; ++q) a_nxt[mi][q] = wp[mi][g + q * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                u32x4 b[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (NOB) {
                        b[q] = bconst[q];
                        asm volatile("" : "+v"(b[q]));
                    } else {
                        b[q] = *reinterpret_cast<const u32x4 *>(cur + q * kPart + bbyte + (ni * 32 + tap * D) * 32);
                    }
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    constexpr int ORD = (FLAGS >> 3) & 3;
                    constexpr int pa[3][6] = {{2, 1, 0, 1, 0, 0}, {0, 0, 0, 1, 1, 2}, {2, 1, 0, 0, 1, 0}};
                    constexpr int pb[3][6] = {{0, 1, 2, 0, 1, 0}, {2, 1, 0, 0, 1, 0}, {0, 0, 0, 1, 1, 2}};
#pragma unroll
                    for (int t = 0; t < 6; ++t)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_cur[mi][pa[ORD][t]]),
                                                                              __builtin_bit_cast(bf16x8, b[pb[ORD][t]]), acc[mi][ni], 0, 0, 0);
                }
            }
            if (!NOA) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int q = 0; q < 3; ++q) a_cur[mi][q] = a_nxt[mi][q];
            } else {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int q = 0; q < 3; ++q) asm volatile("" : "+v"(a_cur[mi][q]));
            }
        }
        if (STAGE && c + 1 < nchunks) stage_store(lds + ((c + 1) & 1) * kBuf);
        if (STAGE) __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[mi][ni][r];
    out[((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x * kThreads + blockIdx.x * kThreads + tid] = s;
    if (tid == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == gridDim.z / 2) {
        g_clk[0] = clock64() - clk0;
        g_clk[1] = wall_clock64() - rt0;
    }
}


// Grouped variant: A is prefetched one GROUP of G taps ahead (register double buffer), and the HBM staging loads of the
// next chunk are issued AFTER the last group's A prefetch, so that no wait on an A load (vmcnt is in-order) ever forces
// the staging loads to have landed before the chunk's final wait.  IL = interleave two n-tiles' MFMA chains.
template <int K, int D, int MI, int NI, int WM, int WN, int OCC, int G, int IL>
__global__ __launch_bounds__(64 * WM * WN, OCC) void x3_grouped(float *out, const u32x4 *__restrict__ wpk, const float *__restrict__ x,
                                                                int nchunks, int x_rstride, int t_in)
{
    constexpr int kThreads = 64 * WM * WN;
    constexpr int kBN = 32 * NI * WN;
    constexpr int kXW = kBN + (K - 1) * D;
    constexpr int kPart = kXW * 32;
    constexpr int kBuf = 3 * kPart;
    constexpr int kItems = 2 * kXW;
    constexpr int kNSt = (kItems + kThreads - 1) / kThreads;
    constexpr int kNG = (K + G - 1) / G;          // groups per chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, h = lane >> 5, j = lane & 31;
    const int t0 = blockIdx.x * kBN;
    const long clk0 = clock64(), rt0 = wall_clock64();

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const u32x4 *wp[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const long mtile = ((long)blockIdx.y * WM + wm) * MI + mi;
        wp[mi] = wpk + mtile * ((long)nchunks * K * 3 * 64) + lane;
    }
    float st[kNSt][8];
    auto stage_load = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < kNSt; ++i) {
            const int e = tid + i * kThreads;
            const int half = e / kXW, col = e - half * kXW;
            const int gt = t0 + col;
            const bool ok = (e < kItems) && gt < t_in;
            const float *src = x + ((long)(chunk * 16 + half * 8)) * x_rstride + gt;
#pragma unroll
            for (int c = 0; c < 8; ++c) st[i][c] = ok ? src[(long)c * x_rstride] : 0.f;
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
                for (int c = 0; c < 8; ++c) {
                    float v = st[i][c];
                    v = v > 0.f ? v : v * 0.1f;
                    split3(v, p[0][c], p[1][c], p[2][c]);
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
    stage_load(0);
    stage_store(lds);
    __syncthreads();

    const int bbyte = (wn * (32 * NI) + j) * 32 + h * 16;
    u32x4 a_cur[G][MI][3], a_nxt[G][MI][3];
#pragma unroll
    for (int t = 0; t < G; ++t)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int q = 0; q < 3; ++q) a_cur[t][mi][q] = wp[mi][(long)t * 3 * 64 + q * 64];

    for (int c = 0; c < nchunks; ++c) {
        const unsigned char *cur = lds + (c & 1) * kBuf;
#pragma unroll
        for (int g = 0; g < kNG; ++g) {
            constexpr int dummy = 0;
            const int tap0 = g * G;
            const int ntap = (K - tap0 < G) ? (K - tap0) : G;
            // next group: taps of this chunk, or the first group of the next chunk (the image has slack at the end)
            const int ntap0 = (g + 1 < kNG) ? (g + 1) * G : K;   // linear tap index relative to chunk c (K = chunk c+1 tap 0)
            const int nn = (g + 1 < kNG) ? ((K - ntap0 < G) ? (K - ntap0) : G) : ((K < G) ? K : G);
#pragma unroll
            for (int t = 0; t < G; ++t)
                if (t < nn) {
                    const long gi = ((long)c * K + ntap0 + t) * 3 * 64;
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int q = 0; q < 3; ++q) a_nxt[t][mi][q] = wp[mi][gi + q * 64];
                }
            if (g == kNG - 1 && c + 1 < nchunks) stage_load(c + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < G; ++t)
                if (t < ntap) {
                    const int tap = tap0 + t;
                    constexpr int pa[6] = {2, 1, 0, 1, 0, 0};
                    constexpr int pb[6] = {0, 1, 2, 0, 1, 0};
                    if (IL && NI % 2 == 0) {
#pragma unroll
                        for (int ni = 0; ni < NI; ni += 2) {
                            u32x4 b0[3], b1[3];
#pragma unroll
                            for (int q = 0; q < 3; ++q) {
                                b0[q] = *reinterpret_cast<const u32x4 *>(cur + q * kPart + bbyte + (ni * 32 + tap * D) * 32);
                                b1[q] = *reinterpret_cast<const u32x4 *>(cur + q * kPart + bbyte + ((ni + 1) * 32 + tap * D) * 32);
                            }
#pragma unroll
                            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                                for (int u = 0; u < 6; ++u) {
                                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_cur[t][mi][pa[u]]),
                                                                                          __builtin_bit_cast(bf16x8, b0[pb[u]]), acc[mi][ni], 0, 0, 0);
                                    acc[mi][ni + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_cur[t][mi][pa[u]]),
                                                                                              __builtin_bit_cast(bf16x8, b1[pb[u]]), acc[mi][ni + 1], 0, 0, 0);
                                }
                        }
                    } else {
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            u32x4 b[3];
#pragma unroll
                            for (int q = 0; q < 3; ++q)
                                b[q] = *reinterpret_cast<const u32x4 *>(cur + q * kPart + bbyte + (ni * 32 + tap * D) * 32);
#pragma unroll
                            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                                for (int u = 0; u < 6; ++u)
                                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_cur[t][mi][pa[u]]),
                                                                                          __builtin_bit_cast(bf16x8, b[pb[u]]), acc[mi][ni], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
            for (int t = 0; t < G; ++t)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int q = 0; q < 3; ++q) a_cur[t][mi][q] = a_nxt[t][mi][q];
        }
        if (c + 1 < nchunks) stage_store(lds + ((c + 1) & 1) * kBuf);
        __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[mi][ni][r];
    out[((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x * kThreads + blockIdx.x * kThreads + tid] = s;
    if (tid == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == gridDim.z / 2) {
        g_clk[0] = clock64() - clk0;
        g_clk[1] = wall_clock64() - rt0;
    }
}

static float *g_out; static u32x4 *g_w; static float *g_x;

template <int K, int D, int MI, int NI, int WM, int WN, int OCC, int FLAGS>
void run(int c, int T, int batch = 32)
{
    constexpr int kBN = 32 * NI * WN, kBM = 32 * MI * WM;
    constexpr int kXW = kBN + (K - 1) * D;
    const size_t ldsb = (size_t)2 * 3 * kXW * 32;
    auto kern = x3_loop<K, D, MI, NI, WM, WN, OCC, FLAGS>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern));
    const int nchunks = c / 16;
    dim3 grid((T + kBN - 1) / kBN, (c + kBM - 1) / kBM, batch);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), ldsb, 0, g_out, g_w, g_x, nchunks, T + 64, T);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double flop = 2.0 * c * c * K * (double)T * batch;
    long hclk[2]; hipMemcpyFromSymbol(hclk, HIP_SYMBOL(g_clk), sizeof(hclk));
    printf("[%.2f GHz] ", hclk[1] ? (double)hclk[0] / hclk[1] * 0.1 : 0.0);
    hipError_t err = hipGetLastError();
    printf("<%d,%d,%d,%d> occ%d flags%d k=%2d C=%3d T=%6d grid=%6d lds=%6zu vgpr=%3d spill=%d: %8.3f ms %7.1f TF-eq (%4.2f of bf16 peak) %s\n",
           MI, NI, WM, WN, OCC, FLAGS, K, c, T, grid.x * grid.y * grid.z, ldsb, fa.numRegs, (int)fa.localSizeBytes, best,
           flop / best / 1e9, 6 * flop / best / 1e9 / 2500.0, err == hipSuccess ? "" : hipGetErrorString(err));
    fflush(stdout);
}


template <int K, int D, int MI, int NI, int WM, int WN, int OCC, int G, int IL>
void rung(int c, int T, int batch = 32)
{
    constexpr int kBN = 32 * NI * WN, kBM = 32 * MI * WM;
    constexpr int kXW = kBN + (K - 1) * D;
    const size_t ldsb = (size_t)2 * 3 * kXW * 32;
    auto kern = x3_grouped<K, D, MI, NI, WM, WN, OCC, G, IL>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern));
    const int nchunks = c / 16;
    dim3 grid((T + kBN - 1) / kBN, (c + kBM - 1) / kBM, batch);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), ldsb, 0, g_out, g_w, g_x, nchunks, T + 64, T);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double flop = 2.0 * c * c * K * (double)T * batch;
    long hclk[2]; hipMemcpyFromSymbol(hclk, HIP_SYMBOL(g_clk), sizeof(hclk));
    printf("[%.2f GHz] ", hclk[1] ? (double)hclk[0] / hclk[1] * 0.1 : 0.0);
    hipError_t err = hipGetLastError();
    printf("GROUPED<%d,%d,%d,%d> occ%d G=%d IL=%d k=%2d C=%3d T=%6d grid=%6d lds=%6zu vgpr=%3d spill=%d: %8.3f ms %7.1f TF-eq (%4.2f of bf16 peak) %s\n",
           MI, NI, WM, WN, OCC, G, IL, K, c, T, grid.x * grid.y * grid.z, ldsb, fa.numRegs, (int)fa.localSizeBytes, best,
           flop / best / 1e9, 6 * flop / best / 1e9 / 2500.0, err == hipSuccess ? "" : hipGetErrorString(err));
    fflush(stdout);
}

template <int K, int G>
void gsuite()
{
    for (int s = 0; s < 2; ++s) {
        const int c = s ? 128 : 256, T = s ? 49280 : 6160;
        run<K, 1, 2, 2, 2, 2, 2, 0>(c, T);
        run<K, 1, 1, 4, 4, 1, 2, 0>(c, T);
        rung<K, 1, 1, 4, 4, 1, 2, G, 0>(c, T);
        rung<K, 1, 1, 4, 4, 1, 2, G, 1>(c, T);
        rung<K, 1, 1, 4, 4, 1, 3, G, 1>(c, T);
        rung<K, 1, 2, 2, 2, 2, 2, G, 0>(c, T);
        rung<K, 1, 2, 2, 2, 2, 2, G, 1>(c, T);
        rung<K, 1, 1, 8, 4, 1, 2, G, 1>(c, T);
        rung<K, 1, 1, 4, 4, 2, 1, G, 1>(c, T);
    }
    run<K, 1, 2, 2, 1, 4, 2, 0>(64, 98560);
    rung<K, 1, 2, 2, 1, 4, 2, G, 0>(64, 98560);
    rung<K, 1, 1, 4, 2, 2, 2, G, 0>(64, 98560);
    rung<K, 1, 1, 4, 2, 2, 2, G, 1>(64, 98560);
    rung<K, 1, 1, 4, 2, 2, 3, G, 1>(64, 98560);
    rung<K, 1, 1, 2, 2, 4, 1, G, 1>(64, 98560);
    rung<K, 1, 1, 2, 2, 4, 2, G, 1>(64, 98560);
    run<K, 1, 1, 2, 1, 4, 2, 0>(32, 197120);
    rung<K, 1, 1, 2, 1, 4, 2, G, 0>(32, 197120);
    rung<K, 1, 1, 2, 1, 4, 2, G, 1>(32, 197120);
    rung<K, 1, 1, 2, 1, 4, 4, G, 1>(32, 197120);
    rung<K, 1, 1, 4, 1, 4, 2, G, 1>(32, 197120);
    rung<K, 1, 1, 2, 1, 8, 2, G, 1>(32, 197120);
}

template <int K>
void suite()
{
    // C=256 / C=128 (128-row block tiles)
    for (int s = 0; s < 2; ++s) {
        const int c = s ? 128 : 256, T = s ? 49280 : 6160;
        run<K, 1, 2, 2, 2, 2, 2, 0>(c, T);
        run<K, 1, 2, 2, 2, 2, 2, 1>(c, T);
        run<K, 1, 2, 2, 2, 2, 2, 2>(c, T);
        run<K, 1, 2, 2, 2, 2, 2, 4>(c, T);
        run<K, 1, 2, 2, 2, 2, 2, 7>(c, T);
        run<K, 1, 1, 4, 4, 1, 2, 0>(c, T);
        run<K, 1, 1, 4, 4, 1, 2, 1>(c, T);
        run<K, 1, 1, 4, 4, 1, 2, 4>(c, T);
        run<K, 1, 1, 4, 4, 1, 3, 0>(c, T);
        run<K, 1, 1, 8, 4, 1, 2, 0>(c, T);
        run<K, 1, 1, 8, 4, 1, 1, 0>(c, T);
        run<K, 1, 2, 4, 2, 2, 1, 0>(c, T);
        run<K, 1, 2, 4, 2, 2, 2, 0>(c, T);
        run<K, 1, 1, 4, 4, 2, 1, 0>(c, T);
        run<K, 1, 2, 4, 2, 1, 2, 0>(c, T);
    }
    run<K, 1, 2, 4, 4, 1, 1, 0>(256, 6160);
    run<K, 1, 2, 4, 4, 1, 2, 0>(256, 6160);
    // C=64
    run<K, 1, 2, 2, 1, 4, 2, 0>(64, 98560);
    run<K, 1, 2, 2, 1, 4, 2, 1>(64, 98560);
    run<K, 1, 2, 2, 1, 4, 2, 4>(64, 98560);
    run<K, 1, 1, 4, 2, 2, 2, 0>(64, 98560);
    run<K, 1, 1, 4, 2, 1, 4, 0>(64, 98560);
    run<K, 1, 1, 8, 2, 1, 2, 0>(64, 98560);
    run<K, 1, 2, 4, 1, 2, 2, 0>(64, 98560);
    // C=32
    run<K, 1, 1, 2, 1, 4, 2, 0>(32, 197120);
    run<K, 1, 1, 2, 1, 4, 2, 4>(32, 197120);
    run<K, 1, 1, 2, 1, 4, 4, 0>(32, 197120);
    run<K, 1, 1, 4, 1, 2, 4, 0>(32, 197120);
    run<K, 1, 1, 4, 1, 4, 2, 0>(32, 197120);
    run<K, 1, 1, 8, 1, 1, 4, 0>(32, 197120);
}

int main(int argc, char **argv)
{
    const size_t wbytes = (size_t)64 << 20, xbytes = (size_t)1 << 30, obytes = (size_t)256 << 20;
    hipMalloc(&g_w, wbytes); hipMalloc(&g_x, xbytes); hipMalloc(&g_out, obytes);
    std::vector<unsigned short> hw(wbytes / 2);
    for (auto &v : hw) v = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
    hipMemcpy(g_w, hw.data(), wbytes, hipMemcpyHostToDevice);
    std::vector<float> hx(xbytes / 4);
    for (auto &v : hx) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(g_x, hx.data(), xbytes, hipMemcpyHostToDevice);
    if (argc > 1 && argv[1][0] == 's') { suite<11>(); suite<3>(); return 0; }
    if (argc > 1 && argv[1][0] == 'o') {
        for (int rep = 0; rep < 3; ++rep) {
            run<11, 1, 2, 2, 2, 2, 2, 0>(128, 49280);
            run<11, 1, 2, 2, 2, 2, 2, 8>(128, 49280);
            run<11, 1, 2, 2, 2, 2, 2, 16>(128, 49280);
            run<11, 1, 2, 2, 2, 2, 2, 7>(128, 49280);
            run<11, 1, 2, 2, 2, 2, 2, 15>(128, 49280);
            run<11, 1, 2, 2, 2, 2, 2, 23>(128, 49280);
            run<3, 1, 2, 2, 2, 2, 2, 0>(128, 49280);
            run<3, 1, 2, 2, 2, 2, 2, 8>(128, 49280);
            run<3, 1, 2, 2, 2, 2, 2, 16>(128, 49280);
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'c') {
        for (int z = 0; z < 2; ++z) {
            if (z) { hipMemset(g_w, 0, wbytes); hipMemset(g_x, 0, xbytes); printf("--- all-zero operands ---\n"); }
            for (int rep = 0; rep < 2; ++rep) {
                run<11, 1, 2, 2, 2, 2, 2, 0>(128, 49280);
                run<11, 1, 2, 2, 2, 2, 2, 7>(128, 49280);
                run<11, 1, 2, 2, 2, 2, 2, 1>(128, 49280);
                run<11, 1, 2, 2, 2, 2, 2, 4>(128, 49280);
                run<11, 1, 1, 4, 4, 1, 2, 0>(128, 49280);
                run<3, 1, 2, 2, 2, 2, 2, 0>(128, 49280);
                run<3, 1, 2, 2, 2, 2, 2, 7>(128, 49280);
                run<11, 1, 1, 2, 1, 4, 2, 0>(32, 197120);
            }
        }
        return 0;
    }
    gsuite<11, 4>();
    gsuite<11, 3>();
    gsuite<7, 4>();
    gsuite<3, 3>();
    return 0;
}
