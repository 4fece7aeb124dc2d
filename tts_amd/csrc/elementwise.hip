// Small streaming kernels: one coalesced pass over HBM each (grid-stride, 256-thread blocks).
#include "common.h"

namespace ttsamd {

constexpr int kEwThreads = 256;
inline int ew_blocks(long n) { long b = (n + kEwThreads - 1) / kEwThreads; return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); }

__global__ void replicate_pad_kernel(float *__restrict__ y, const float *__restrict__ x, long rows, int t, int pad)
{
    const int to = t + 2 * pad;
    const long n = rows * to;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / to;
        int c = (int)(i - r * to) - pad;
        c = c < 0 ? 0 : (c >= t ? t - 1 : c);
        y[i] = x[r * t + c];
    }
}

}  // namespace ttsamd
using namespace ttsamd;

extern "C" int ttsamd_replicate_pad(float *y, const float *x, int64_t rows, int t, int pad, void *stream)
{
    TTSAMD_CHECK_ARG(y && x && rows >= 0 && t > 0 && pad >= 0, "replicate_pad: bad args");
    if (rows == 0) return TTSAMD_OK;
    hipLaunchKernelGGL(replicate_pad_kernel, dim3(ew_blocks(rows * (t + 2 * pad))), dim3(kEwThreads), 0, as_stream(stream), y, x, (long)rows, t, pad);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}
