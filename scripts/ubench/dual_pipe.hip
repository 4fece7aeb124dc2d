// Can the fp32 MFMA pipe and the fp32 VALU pipe run flat out at the same time?  Per block: waves 0-3 issue only
// v_mfma_f32_32x32x2_f32, waves 4-7 only v_fma_f32 (register operands, random data).  Reports both rates when run
// alone and together.   hipcc --offload-arch=gfx950 -O3 dual_pipe.hip -o dual_pipe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ __launch_bounds__(512) void k(float *out, const float *in, int iters, int mode /*1 mfma,2 valu,3 both*/)
{
    const int wave = threadIdx.x >> 6;
    const bool is_mfma = wave < 4;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 4095]; b[i] = in[(threadIdx.x * 8 + i + 977) & 4095]; }
    float s = 0.f;
    if (is_mfma) {
        if (!(mode & 1)) return;
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[(q + i) & 7], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        if (!(mode & 2)) return;
        float acc[64];
        for (int i = 0; i < 64; ++i) acc[i] = 0.f;
        // the same FLOP count per iteration as the MFMA waves: 32 MFMA * 4096 flop = 131072 flop per wave-iteration
        // = 1024 wave-level v_fma (128 flop each) -> 16 passes over 64 accumulators
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int p = 0; p < 16; ++p)
#pragma unroll
                for (int i = 0; i < 64; ++i) acc[i] = __builtin_fmaf(a[(i + p) & 7], b[p & 7], acc[i]);
        }
        for (int i = 0; i < 64; ++i) s += acc[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    float *in, *out;
    (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&out, 4096 * 512 * 4);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000, grid = 256;
    for (int mode = 1; mode <= 3; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, out, in, iters, mode);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double per_kind = (double)grid * 4 * iters * 131072.0;
            const double flops = per_kind * ((mode & 1 ? 1 : 0) + (mode & 2 ? 1 : 0));
            printf("mode=%d (%s) rep=%d: %.2f ms  %.1f TFLOP/s total\n", mode, mode == 1 ? "mfma only" : mode == 2 ? "valu only" : "both", rep, ms, flops / ms / 1e9);
        }
    return 0;
}
