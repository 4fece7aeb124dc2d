// fp32-input MFMA peak micro-benchmark (register operands only).  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 4095]; b[i] = in[(threadIdx.x * 8 + i + 977) & 4095]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[(s + i) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char **argv)
{
    const int zero = argc > 1 ? atoi(argv[1]) : 0;
    const int nacc = argc > 2 ? atoi(argv[2]) : 4;
    float *in, *out;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 4096 * 256 * 4);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = zero ? 0.f : (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
        const int grid = 256 * blocks_per_cu;
        for (int rep = 0; rep < 4; ++rep) {
            const int iters = 20000;
            hipEventRecord(e0);
            if (nacc == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, in, iters);
            else if (nacc == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, in, iters);
            else hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, in, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)grid * 4 * iters * 8 * nacc * 4096.0;
            printf("nacc=%d zero=%d blocks/CU=%d rep=%d: %.2f ms  %.1f TFLOP/s\n", nacc, zero, blocks_per_cu, rep, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
