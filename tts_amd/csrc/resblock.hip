// Dispatcher of the fused ResBlock1-iteration kernel (resblock_kernel_x3.h).
#include "resblock_kernel_x3.h"

using namespace ttsamd;

extern "C" int ttsamd_resblock_pair_supported(int c, int kernel, int dilation)
{
    return (c == 8 || c == 16 || c == 32 || c == 64 || c == 128) && (kernel == 3 || kernel == 7 || kernel == 11) &&
           (dilation == 1 || dilation == 3 || dilation == 5);
}

extern "C" size_t ttsamd_resblock_weight_bytes(int c, int kernel)
{
    const int cc = c < 32 ? 32 : c;
    return ttsamd_conv1d_packed_split_bytes(cc, cc, kernel);
}

extern "C" int ttsamd_resblock_pair(const ttsamd_resblock_args *args, void *stream)
{
    TTSAMD_CHECK_ARG(args, "resblock_pair: NULL args");
    const ttsamd_resblock_args &a = *args;
    TTSAMD_CHECK_ARG(a.x && a.y && a.w1_split && a.w2_split, "resblock_pair: NULL tensor");
    TTSAMD_CHECK_ARG(a.x != a.y, "resblock_pair: y must not alias x (neighbouring tiles read x's halo)");
    TTSAMD_CHECK_ARG(a.c > 0 && a.t >= 0 && a.batch >= 0, "resblock_pair: bad shape");
    TTSAMD_CHECK_ARG(a.slope >= 0.f && a.slope <= 1.f, "resblock_pair: leaky-ReLU slope %g outside [0, 1]", (double)a.slope);
    if (!ttsamd_resblock_pair_supported(a.c, a.kernel, a.dilation)) {
        set_error("resblock_pair: (c=%d, kernel=%d, dilation=%d) has no instantiation", a.c, a.kernel, a.dilation);
        return TTSAMD_ERR_UNSUPPORTED;
    }
    {
        const int64_t need = (int64_t)ttsamd_resblock_weight_bytes(a.c, a.kernel);
        TTSAMD_CHECK_ARG(a.w1_bytes == need && a.w2_bytes == need,
                         "resblock_pair: weight images of %lld / %lld bytes, the c=%d k=%d tile reads %lld (c < 32: pack the "
                         "weight zero-padded to [32, 32, k])", (long long)a.w1_bytes, (long long)a.w2_bytes, a.c, a.kernel,
                         (long long)need);
    }
    if (a.batch == 0 || a.t == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(a.batch <= 65535, "resblock_pair: batch > 65535");
    if ((int64_t)a.c * a.t * 4 >= 0x7FFFFFF0) {
        set_error("resblock_pair: a per-item [C, T] slab exceeds 2 GiB (time-tile the call)");
        return TTSAMD_ERR_UNSUPPORTED;
    }
    hipStream_t st = as_stream(stream);
    switch (a.kernel) {
        case 3: return resblock_pair_launch_k3(a, st);
        case 7: return resblock_pair_launch_k7(a, st);
        case 11: return resblock_pair_launch_k11(a, st);
    }
    return TTSAMD_ERR_UNSUPPORTED;
}

