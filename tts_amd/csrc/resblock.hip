// Dispatcher of the fused ResBlock1-iteration kernel (resblock_kernel_x3.h).
#include "resblock_kernel_x3.h"

using namespace ttsamd;

extern "C" int ttsamd_resblock_pair(const ttsamd_resblock_args *args, void *stream)
{
    TTSAMD_CHECK_ARG(args, "resblock_pair: NULL args");
    const ttsamd_resblock_args &a = *args;
    TTSAMD_CHECK_ARG(a.x && a.y && a.w1_split && a.w2_split, "resblock_pair: NULL tensor");
    TTSAMD_CHECK_ARG(a.x != a.y, "resblock_pair: y must not alias x (neighbouring tiles read x's halo)");
    TTSAMD_CHECK_ARG(a.c > 0 && a.t >= 0 && a.batch >= 0, "resblock_pair: bad shape");
    TTSAMD_CHECK_ARG(a.slope >= 0.f && a.slope <= 1.f, "resblock_pair: leaky-ReLU slope %g outside [0, 1]", (double)a.slope);
    if (!ttsamd_resblock_pair_supported(a.c, a.kernel, a.dilation) &&
        !(a.w1_h2 && a.w2_h2 && a.variant != 2 && ttsamd_resblock_pair_h2_supported(a.c, a.kernel, a.dilation))) {
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
    if (a.w1_h2 || a.w2_h2) {
        const int64_t need = (int64_t)ttsamd_resblock_weight_h2_bytes(a.c, a.kernel);
        TTSAMD_CHECK_ARG(a.w1_h2 && a.w2_h2 && a.w1_h2_bytes == need && a.w2_h2_bytes == need,
                         "resblock_pair: two-part fp16 images of %lld / %lld bytes, the c=%d k=%d tile reads %lld (both or neither)",
                         (long long)a.w1_h2_bytes, (long long)a.w2_h2_bytes, a.c, a.kernel, (long long)need);
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


extern "C" int ttsamd_resblock_group_supported(int c, int t, int batch)
{
    return (c == 8 || c == 16 || c == 32 || c == 64) && resblock_group_small(c, (long)t * batch);
}

extern "C" int ttsamd_resblock_group(const ttsamd_resblock_args *args3, void *stream)
{
    TTSAMD_CHECK_ARG(args3, "resblock_group: NULL args");
    static const int kSlotKernel[3] = {3, 7, 11};
    ResGroupArgs g;
    const ttsamd_resblock_args *first = nullptr;
    for (int i = 0; i < 3; ++i) {
        g.br[i] = args3[i];
        const ttsamd_resblock_args &a = g.br[i];
        if (!a.x) continue;
        TTSAMD_CHECK_ARG(a.y && a.w1_split && a.w2_split, "resblock_group: NULL tensor in slot %d", i);
        TTSAMD_CHECK_ARG(a.x != a.y, "resblock_group: y must not alias x (slot %d)", i);
        TTSAMD_CHECK_ARG(a.kernel == kSlotKernel[i], "resblock_group: slot %d takes kernel size %d, not %d", i, kSlotKernel[i], a.kernel);
        TTSAMD_CHECK_ARG(a.slope >= 0.f && a.slope <= 1.f, "resblock_group: leaky-ReLU slope %g outside [0, 1]", (double)a.slope);
        const int64_t need = (int64_t)ttsamd_resblock_weight_bytes(a.c, a.kernel);
        TTSAMD_CHECK_ARG(a.w1_bytes == need && a.w2_bytes == need, "resblock_group: slot %d weight images of %lld / %lld bytes, the c=%d k=%d tile reads %lld",
                         i, (long long)a.w1_bytes, (long long)a.w2_bytes, a.c, a.kernel, (long long)need);
        if (!first) first = &a;
        TTSAMD_CHECK_ARG(a.c == first->c && a.t == first->t && a.batch == first->batch && a.dilation == first->dilation,
                         "resblock_group: the branches of one launch share c, t, batch and dilation");
    }
    TTSAMD_CHECK_ARG(first, "resblock_group: no branch");
    TTSAMD_CHECK_ARG(first->c > 0 && first->t >= 0 && first->batch >= 0, "resblock_group: bad shape");
    if (first->batch == 0 || first->t == 0) return TTSAMD_OK;
    if (!ttsamd_resblock_group_supported(first->c, first->t, first->batch)) {
        set_error("resblock_group: c=%d, t=%d, batch=%d is outside the grouped range (ttsamd_resblock_group_supported): launch the pairs one by one",
                  first->c, first->t, first->batch);
        return TTSAMD_ERR_UNSUPPORTED;
    }
    TTSAMD_CHECK_ARG(first->batch <= 65535, "resblock_group: batch > 65535");
    hipStream_t st = as_stream(stream);
    switch (first->dilation) {
        case 1: return resblock_group_launch_d1(g, first->c, first->t, first->batch, st);
        case 3: return resblock_group_launch_d3(g, first->c, first->t, first->batch, st);
        case 5: return resblock_group_launch_d5(g, first->c, first->t, first->batch, st);
    }
    set_error("resblock_group: dilation %d has no instantiation (1, 3, 5)", first->dilation);
    return TTSAMD_ERR_UNSUPPORTED;
}
