// Prototype of a conv main loop on the bf16 MFMA with a 3-way bf16 split of both fp32 operands (6 products per fp32
// product, fp32 accumulate: error below one fp32 rounding of the product).  Measures what the loop structure can
// reach before the real kernel is written:  hipcc --offload-arch=gfx950 -O3 x3_loop.hip -o x3_loop
//   B (activations): LDS strip [part][col][16 ch] bf16, fragments = ds_read_b128 (lane -> col, half-wave -> 8 channels)
//   A (weights): pre-split, pre-packed [mtile][chunk][tap][part][64 lanes][8 bf16], b128 loads from L2
//   optional staging: fp32 [16 ch][XW] from global -> split -> LDS, double buffered (as the real kernel would)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

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

template <int K, int D, int MI, int NI, int WM, int WN, int OCC, bool STAGE>
__global__ __launch_bounds__(64 * WM * WN, OCC) void x3_loop(float *out, const u32x4 *__restrict__ wpk, const float *__restrict__ x,
                                                             int nchunks, int x_rstride, int t_in)
{
    constexpr int kThreads = 64 * WM * WN;
    constexpr int kBN = 32 * NI * WN;
    constexpr int kXW = kBN + (K - 1) * D;
    constexpr int kPart = kXW * 32;              // bytes per part per buffer
    constexpr int kBuf = 3 * kPart;
    constexpr int kItems = 2 * kXW;              // (col, half) work items per chunk
    constexpr int kNSt = (kItems + kThreads - 1) / kThreads;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, h = lane >> 5, j = lane & 31;
    const int t0 = blockIdx.x * kBN;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // A stream: per m-tile [chunk][tap][part][64 lanes] uint4
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

    for (int c = 0; c < nchunks; ++c) {
        const unsigned char *cur = lds + (STAGE ? (c & 1) * kBuf : 0);
        if (STAGE && c + 1 < nchunks) stage_load(c + 1);
#pragma unroll
        for (int tap = 0; tap < K; ++tap) {
            const long g = ((long)c * K + tap + 1) * 3 * 64;   // next tap's A (one slack group at the end of the image)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int q = 0; q < 3; ++q) a_nxt[mi][q] = wp[mi][g + q * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                u32x4 b[3];
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    b[q] = *reinterpret_cast<const u32x4 *>(cur + q * kPart + bbyte + (ni * 32 + tap * D) * 32);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    // smallest terms first
                    constexpr int pa[6] = {2, 1, 0, 1, 0, 0};
                    constexpr int pb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                    for (int t = 0; t < 6; ++t)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_cur[mi][pa[t]]),
                                                                              __builtin_bit_cast(bf16x8, b[pb[t]]), acc[mi][ni], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int q = 0; q < 3; ++q) a_cur[mi][q] = a_nxt[mi][q];
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
}

template <int K, int D, int MI, int NI, int WM, int WN, int OCC, bool STAGE>
void run(const char *name, int c_in, int c_out, int T, int batch, float *out, const u32x4 *w, const float *x)
{
    constexpr int kBN = 32 * NI * WN, kBM = 32 * MI * WM;
    constexpr int kXW = kBN + (K - 1) * D;
    const size_t ldsb = (size_t)2 * 3 * kXW * 32;
    auto kern = x3_loop<K, D, MI, NI, WM, WN, OCC, STAGE>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    const int nchunks = c_in / 16;
    dim3 grid((T + kBN - 1) / kBN, c_out / kBM, batch);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), ldsb, 0, out, w, x, nchunks, T + 64, T);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double flop = 2.0 * c_out * c_in * K * (double)grid.x * kBN * batch;
    hipError_t err = hipGetLastError();
    printf("%-34s k=%2d C=%3d->%3d T=%6d B=%2d grid=%5d lds=%6zu: %8.3f ms  %7.1f TF-equiv (x6 = %6.0f TF bf16)  %s\n", name, K, c_in, c_out, T,
           batch, grid.x * grid.y * grid.z, ldsb, best, flop / best / 1e9, 6 * flop / best / 1e9, err == hipSuccess ? "" : hipGetErrorString(err));
}

int main()
{
    const size_t wbytes = (size_t)64 << 20, xbytes = (size_t)1 << 30, obytes = (size_t)256 << 20;
    u32x4 *w; float *x, *out;
    hipMalloc(&w, wbytes); hipMalloc(&x, xbytes); hipMalloc(&out, obytes);
    std::vector<unsigned short> hw(wbytes / 2);
    for (auto &v : hw) v = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);   // bf16 around +-0.01
    hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice);
    std::vector<float> hx(xbytes / 4);
    for (auto &v : hx) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(x, hx.data(), xbytes, hipMemcpyHostToDevice);
    // VITS decoder stage 1 (C=256, T=6160, B=32), the dominant k=11 layer, and k=3
    run<11, 1, 2, 4, 2, 2, 1, false>("128x256 occ1 no-stage", 256, 256, 6160, 32, out, w, x);
    run<11, 1, 2, 4, 2, 2, 1, true>("128x256 occ1 stage", 256, 256, 6160, 32, out, w, x);
    run<11, 1, 2, 2, 2, 2, 2, false>("128x128 occ2 no-stage", 256, 256, 6160, 32, out, w, x);
    run<11, 1, 2, 2, 2, 2, 2, true>("128x128 occ2 stage", 256, 256, 6160, 32, out, w, x);
    run<11, 1, 2, 2, 2, 2, 1, true>("128x128 occ1 stage", 256, 256, 6160, 32, out, w, x);
    run<3, 1, 2, 4, 2, 2, 1, true>("128x256 occ1 stage", 256, 256, 6160, 32, out, w, x);
    run<3, 1, 2, 2, 2, 2, 2, true>("128x128 occ2 stage", 256, 256, 6160, 32, out, w, x);
    run<11, 5, 2, 4, 2, 2, 1, true>("128x256 occ1 stage d5", 256, 256, 6160, 32, out, w, x);
    // stage 2 (C=128, T=49280)
    run<11, 1, 2, 4, 2, 2, 1, true>("128x256 occ1 stage", 128, 128, 49280, 32, out, w, x);
    run<11, 1, 2, 2, 2, 2, 2, true>("128x128 occ2 stage", 128, 128, 49280, 32, out, w, x);
    return 0;
}
