// Bitwise check of the fma_mix form of the hi / lo split against the convert-back form (conv_split2x2), 4 M random pairs with
// exponents over fp16 normal / denormal / underflow:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off split_mix.hip -o split_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
using f16x2v = __attribute__((ext_vector_type(2))) _Float16;
using f32x2v = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ void split_old(float x0, float x1, unsigned &whi, unsigned &wlo)
{
    const f32x2v v = {x0, x1};
    const f16x2v hi = __builtin_convertvector(v, f16x2v);
    const f32x2v hf = __builtin_convertvector(hi, f32x2v);
    const f32x2v r = (v - hf) * 2048.f;
    whi = __builtin_bit_cast(unsigned, hi);
    wlo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2v));
}
__device__ __forceinline__ void split_new(float x0, float x1, unsigned &whi, unsigned &wlo)
{
    const f32x2v v = {x0, x1};
    whi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2v));
    const f32x2v v2k = v * 2048.f;
    const float m = -2048.f;
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(whi), "s"(m), "v"(v2k[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(whi), "s"(m), "v"(v2k[1]));
    wlo = lo;
}
__global__ void k(const float2 *in, uint4 *out, float s, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float2 x = in[i];
    unsigned a, b, c, d;
    split_old(x.x * s, x.y * s, a, b);
    split_new(x.x * s, x.y * s, c, d);
    out[i] = make_uint4(a, b, c, d);
}
int main()
{
    const int n = 1 << 22;
    float2 *h = (float2 *)malloc(n * sizeof(float2));
    srand(1);
    for (int i = 0; i < n; ++i) {
        unsigned u0 = ((unsigned)rand() << 16) ^ (unsigned)rand(), u1 = ((unsigned)rand() << 16) ^ (unsigned)rand();
        // exponent in a range that covers fp16 normal, denormal, underflow, and up to 2^15
        int e0 = 127 - 40 + rand() % 56, e1 = 127 - 40 + rand() % 56;
        u0 = (u0 & 0x807FFFFFu) | ((unsigned)e0 << 23); u1 = (u1 & 0x807FFFFFu) | ((unsigned)e1 << 23);
        memcpy(&h[i].x, &u0, 4); memcpy(&h[i].y, &u1, 4);
    }
    h[0] = make_float2(0.f, -0.f); h[1] = make_float2(65504.f, -65520.f); h[2] = make_float2(1e-30f, 6e-8f);
    float2 *din; uint4 *dout; hipMalloc(&din, n * sizeof(float2)); hipMalloc(&dout, n * sizeof(uint4));
    hipMemcpy(din, h, n * sizeof(float2), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, din, dout, 1.0f, n);
    uint4 *o = (uint4 *)malloc(n * sizeof(uint4));
    hipMemcpy(o, dout, n * sizeof(uint4), hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i) if (o[i].x != o[i].z || o[i].y != o[i].w) { if (bad < 10) printf("diff at %d: x=(%g,%g) old %08x %08x new %08x %08x\n", i, h[i].x, h[i].y, o[i].x, o[i].y, o[i].z, o[i].w); ++bad; }
    printf("mismatches: %ld of %d\n", bad, n);
    return bad != 0;
}
