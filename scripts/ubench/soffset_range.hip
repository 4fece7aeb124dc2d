// Does the hardware range check of a raw buffer access on gfx950 include the SCALAR offset?  (round 6: the staging loads were briefly
// written with their row offset in soffset.)  rsrc covers the first 64 bytes of a 256-byte array holding 1, 2, 3, ...
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(float *data, float *out)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(data, 0, 64, 0x00020000);
    const int lane = threadIdx.x;
    if (lane == 0) {
        out[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 0, 0, 0));        // in range: 1
        out[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 128, 0, 0));      // voffset beyond: 0
        out[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 0, 128, 0));      // soffset beyond: 0 if checked, 33 if not
        out[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 32, 64, 0));      // sum beyond, each in range: 0 if the SUM is checked, 25 if not
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, -7.f), r, 0, 192, 0);     // store with soffset beyond: dropped if checked
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, -9.f), r, 200, 0, 0);     // store with voffset beyond: dropped
    }
}
int main()
{
    float h[64], o[4] = {-1, -1, -1, -1}, *d, *dout;
    for (int i = 0; i < 64; ++i) h[i] = (float)(i + 1);
    hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice); hipMemcpy(dout, o, sizeof(o), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, dout);
    hipDeviceSynchronize();
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("load in range %.0f | voffset beyond %.0f | soffset beyond %.0f (0 = range-checked, 33 = not) | voffset + soffset beyond %.0f (0 = sum checked, 25 = not)\n", o[0], o[1], o[2], o[3]);
    printf("store soffset beyond: data[48] = %.0f (49 = dropped, -7 = written) | store voffset beyond: data[50] = %.0f (51 = dropped)\n", h[48], h[50]);
    return 0;
}
