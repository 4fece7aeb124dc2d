// Small streaming kernels: one coalesced pass over HBM each (grid-stride, 256-thread blocks).
#include "common.h"

namespace ttsamd {

constexpr int kEwThreads = 256;
inline int ew_blocks(long n) { long b = (n + kEwThreads - 1) / kEwThreads; return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); }

__global__ void replicate_pad_kernel(float *__restrict__ y, const float *__restrict__ x, long rows, int t, int pad,
                                     const long *__restrict__ lengths, int rows_per_item, long len_bias = 0)
{
    const int to = t + 2 * pad;
    const long n = rows * to;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / to;
        int c = (int)(i - r * to) - pad;
        int last = t - 1;
        if (lengths) {  // ragged batch: every item replicates ITS OWN last valid frame
            const long l = lengths[r / rows_per_item] + len_bias;
            last = (int)(l < 1 ? 0 : (l > t ? t - 1 : l - 1));
        }
        c = c < 0 ? 0 : (c > last ? last : c);
        y[i] = x[r * t + c];
    }
}

}  // namespace ttsamd
using namespace ttsamd;

extern "C" int ttsamd_replicate_pad(float *y, const float *x, int64_t rows, int t, int pad, void *stream)
{
    TTSAMD_CHECK_ARG(y && x && rows >= 0 && t > 0 && pad >= 0, "replicate_pad: bad args");
    if (rows == 0) return TTSAMD_OK;
    hipLaunchKernelGGL(replicate_pad_kernel, dim3(ew_blocks(rows * (t + 2 * pad))), dim3(kEwThreads), 0, as_stream(stream), y, x, (long)rows, t, pad,
                       (const long *)nullptr, 1);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_replicate_pad_ragged(float *y, const float *x, const int64_t *lengths, int batch, int c, int t,
                                           int pad, void *stream)
{
    TTSAMD_CHECK_ARG(y && x && lengths && batch >= 0 && c > 0 && t > 0 && pad >= 0, "replicate_pad_ragged: bad args");
    if (batch == 0) return TTSAMD_OK;
    const long rows = (long)batch * c;
    hipLaunchKernelGGL(replicate_pad_kernel, dim3(ew_blocks(rows * (t + 2 * pad))), dim3(kEwThreads), 0, as_stream(stream), y, x, rows, t, pad,
                       reinterpret_cast<const long *>(lengths), c);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_replicate_pad_ragged_ex(float *y, const float *x, const int64_t *lengths, int64_t len_bias, int batch,
                                              int c, int t, int pad, void *stream)
{
    TTSAMD_CHECK_ARG(y && x && lengths && batch >= 0 && c > 0 && t > 0 && pad >= 0, "replicate_pad_ragged_ex: bad args");
    if (batch == 0) return TTSAMD_OK;
    const long rows = (long)batch * c;
    hipLaunchKernelGGL(replicate_pad_kernel, dim3(ew_blocks(rows * (t + 2 * pad))), dim3(kEwThreads), 0, as_stream(stream), y, x, rows, t, pad,
                       reinterpret_cast<const long *>(lengths), c, (long)len_bias);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

// ---- Synthesizer mel seam on the device --------------------------------------------------------
namespace ttsamd {

__device__ __forceinline__ float mel_denorm(float s, int c, const ttsamd_mel_norm &p)
{
    if (!p.signal_norm) return s;
    if (p.mean) return s * p.scale[c] + p.mean[c];                      // StandardScaler.inverse_transform
    if (p.symmetric_norm) {
        if (p.clip_norm) s = fminf(fmaxf(s, -p.max_norm), p.max_norm);
        s = ((s + p.max_norm) * -p.min_level_db / (2.f * p.max_norm)) + p.min_level_db;
    } else {
        if (p.clip_norm) s = fminf(fmaxf(s, 0.f), p.max_norm);
        s = (s * -p.min_level_db / p.max_norm) + p.min_level_db;
    }
    return s + p.ref_level_db;
}

__device__ __forceinline__ float mel_norm(float s, int c, const ttsamd_mel_norm &p)
{
    if (!p.signal_norm) return s;
    if (p.mean) return (s - p.mean[c]) / p.scale[c];                    // StandardScaler.transform
    s -= p.ref_level_db;
    float n = (s - p.min_level_db) / (-p.min_level_db);
    if (p.symmetric_norm) {
        n = ((2.f * p.max_norm) * n) - p.max_norm;
        if (p.clip_norm) n = fminf(fmaxf(n, -p.max_norm), p.max_norm);
    } else {
        n = p.max_norm * n;
        if (p.clip_norm) n = fminf(fmaxf(n, 0.f), p.max_norm);
    }
    return n;
}

__global__ void mel_renorm_kernel(float *__restrict__ y, const float *__restrict__ x, ttsamd_mel_norm tts,
                                  ttsamd_mel_norm voc, int C, long T, long total)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)((i / T) % C);
        y[i] = mel_norm(mel_denorm(x[i], c, tts), c, voc);
    }
}

}  // namespace ttsamd

extern "C" int ttsamd_mel_renorm(float *y, const float *x, const ttsamd_mel_norm *tts, const ttsamd_mel_norm *voc,
                                 int batch, int c, int t, void *stream)
{
    TTSAMD_CHECK_ARG(y && x && tts && voc && batch >= 0 && c > 0 && t >= 0, "mel_renorm: bad args");
    TTSAMD_CHECK_ARG((tts->mean == nullptr) == (tts->scale == nullptr) && (voc->mean == nullptr) == (voc->scale == nullptr),
                     "mel_renorm: mean and scale go together");
    const long total = (long)batch * c * t;
    if (total == 0) return TTSAMD_OK;
    hipLaunchKernelGGL(mel_renorm_kernel, dim3(ew_blocks(total)), dim3(kEwThreads), 0, as_stream(stream), y, x, *tts, *voc,
                       c, (long)t, total);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

// ---- speaker conditioning helpers ------------------------------------------------------------------------------
namespace ttsamd {

// y[r,:] = x[r,:] / max(||x[r,:]||_2, eps)   (F.normalize, vits.py:882); one wavefront per row
__global__ void l2_normalize_kernel(float *__restrict__ y, const float *__restrict__ x, int rows, int cols, float eps)
{
    const int r = blockIdx.x;
    const int lane = threadIdx.x;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) { const float v = x[(long)r * cols + c]; s += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float d = fmaxf(sqrtf(s), eps);
    for (int c = lane; c < cols; c += 64) y[(long)r * cols + c] = x[(long)r * cols + c] / d;
}

// y[b,c,t] = x[b,c,t] + rb[b,c]     (DurationPredictor: x + cond(g), duration_predictor.py:58-59)
__global__ void add_row_bias_kernel(float *__restrict__ y, const float *__restrict__ x, const float *__restrict__ rb,
                                    long rows, int t)
{
    const long n = rows * t;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = x[i] + rb[i / t];
}

}  // namespace ttsamd

extern "C" int ttsamd_l2_normalize(float *y, const float *x, int rows, int cols, float eps, void *stream)
{
    TTSAMD_CHECK_ARG(y && x && rows >= 0 && cols > 0, "l2_normalize: bad args");
    if (rows == 0) return TTSAMD_OK;
    hipLaunchKernelGGL(l2_normalize_kernel, dim3(rows), dim3(64), 0, as_stream(stream), y, x, rows, cols, eps);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_add_row_bias(float *y, const float *x, const float *row_bias, int64_t rows, int t, void *stream)
{
    TTSAMD_CHECK_ARG(y && x && row_bias && rows >= 0 && t >= 0, "add_row_bias: bad args");
    if (rows == 0 || t == 0) return TTSAMD_OK;
    hipLaunchKernelGGL(add_row_bias_kernel, dim3(ew_blocks(rows * t)), dim3(kEwThreads), 0, as_stream(stream), y, x, row_bias,
                       (long)rows, t);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

// ---- linear interpolation along time (F.interpolate(mode="linear", align_corners=False, scale_factor=s)) ---------
namespace ttsamd {
__global__ void linear_interp_kernel(float *__restrict__ y, const float *__restrict__ x, long rows, int t_in, int t_out,
                                     float rscale)
{
    const long n = rows * t_out;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / t_out;
        const int o = (int)(i - r * t_out);
        float src = rscale * ((float)o + 0.5f) - 0.5f;        // area_pixel_compute_source_index (align_corners=False)
        src = src < 0.f ? 0.f : src;
        const int i0 = min((int)src, t_in - 1);
        const int i1 = i0 + (i0 < t_in - 1 ? 1 : 0);
        const float l1 = src - (float)i0, l0 = 1.f - l1;
        y[i] = l0 * x[r * t_in + i0] + l1 * x[r * t_in + i1];
    }
}
}  // namespace ttsamd

extern "C" int ttsamd_linear_interp(float *y, const float *x, int64_t rows, int t_in, int t_out, double scale_factor,
                                    void *stream)
{
    TTSAMD_CHECK_ARG(y && x && rows >= 0 && t_in > 0 && t_out >= 0 && scale_factor > 0, "linear_interp: bad args");
    if (rows == 0 || t_out == 0) return TTSAMD_OK;
    hipLaunchKernelGGL(linear_interp_kernel, dim3(ew_blocks(rows * t_out)), dim3(kEwThreads), 0, as_stream(stream), y, x,
                       (long)rows, t_in, t_out, (float)(1.0 / scale_factor));
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

// ---- request glue: several strided copies in ONE launch; all stage masks of a ragged vocoder call in ONE launch ----------
namespace ttsamd {
struct CopyBatch {
    ttsamd_copy_seg seg[TTSAMD_COPY_MAX_SEGS];
};

__global__ void copy_strided_kernel(CopyBatch cb)
{
    const ttsamd_copy_seg &g = cb.seg[blockIdx.y];
    const long n = (long)g.d0 * g.d1 * g.d2;
    const long plane = (long)g.d1 * g.d2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long a = i / plane;
        const long r = i - a * plane;
        const long b = r / g.d2;
        const long c = r - b * g.d2;
        const long so = a * g.s0 + b * g.s1 + c * g.s2;
        const long to = a * g.t0 + b * g.t1 + c * g.t2;
        if (g.elem_bytes == 8)
            reinterpret_cast<unsigned long long *>(g.dst)[to] = reinterpret_cast<const unsigned long long *>(g.src)[so];
        else
            reinterpret_cast<unsigned *>(g.dst)[to] = reinterpret_cast<const unsigned *>(g.src)[so];
    }
}

struct StageMasks {
    int scale[TTSAMD_MASK_MAX_STAGES];
    int t[TTSAMD_MASK_MAX_STAGES];
    long off[TTSAMD_MASK_MAX_STAGES];     // float offset of stage s's [batch, t[s]] mask in `masks`
};

__global__ void stage_masks_kernel(float *__restrict__ masks, long *__restrict__ len_out, const long *__restrict__ lengths,
                                   int batch, int quantum, int add, StageMasks sm)
{
    const int s = blockIdx.y;
    const int t = sm.t[s];
    const long n = (long)batch * t;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long b = i / t;
        const long c = i - b * t;
        const long len = lengths[b] / quantum * quantum + add;
        masks[sm.off[s] + i] = (c < len * sm.scale[s]) ? 1.f : 0.f;
        if (s == 0 && c == 0 && len_out) len_out[b] = len;
    }
}
}  // namespace ttsamd

extern "C" int ttsamd_copy_strided(const ttsamd_copy_seg *segs, int n, void *stream)
{
    TTSAMD_CHECK_ARG(segs && n >= 0 && n <= TTSAMD_COPY_MAX_SEGS, "copy_strided: 0..%d segments", TTSAMD_COPY_MAX_SEGS);
    if (n == 0) return TTSAMD_OK;
    CopyBatch cb;
    long most = 0;
    for (int i = 0; i < n; ++i) {
        const ttsamd_copy_seg &g = segs[i];
        TTSAMD_CHECK_ARG((g.elem_bytes == 4 || g.elem_bytes == 8) && g.d0 >= 0 && g.d1 >= 0 && g.d2 >= 0,
                         "copy_strided: segment %d: element size 4 or 8, non-negative extents", i);
        const long cnt = (long)g.d0 * g.d1 * g.d2;
        TTSAMD_CHECK_ARG(cnt == 0 || (g.src && g.dst), "copy_strided: segment %d: NULL pointer", i);
        cb.seg[i] = g;
        most = cnt > most ? cnt : most;
    }
    if (most == 0) return TTSAMD_OK;
    hipLaunchKernelGGL(copy_strided_kernel, dim3(ew_blocks(most), n), dim3(kEwThreads), 0, as_stream(stream), cb);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_stage_masks(float *masks, int64_t *len_out, const int64_t *lengths, int batch, int quantum, int add,
                                  const int32_t *scales, const int32_t *t_stage, int n_stages, void *stream)
{
    TTSAMD_CHECK_ARG(masks && lengths && scales && t_stage && batch >= 0 && quantum >= 1 && n_stages >= 1 &&
                         n_stages <= TTSAMD_MASK_MAX_STAGES,
                     "stage_masks: bad args (1..%d stages, quantum >= 1)", TTSAMD_MASK_MAX_STAGES);
    if (batch == 0) return TTSAMD_OK;
    StageMasks sm;
    long off = 0, most = 0;
    for (int s = 0; s < n_stages; ++s) {
        TTSAMD_CHECK_ARG(scales[s] >= 1 && t_stage[s] >= 1, "stage_masks: stage %d: scale and length must be >= 1", s);
        sm.scale[s] = scales[s];
        sm.t[s] = t_stage[s];
        sm.off[s] = off;
        off += (long)batch * t_stage[s];
        most = (long)batch * t_stage[s] > most ? (long)batch * t_stage[s] : most;
    }
    hipLaunchKernelGGL(stage_masks_kernel, dim3(ew_blocks(most), n_stages), dim3(kEwThreads), 0, as_stream(stream), masks,
                       reinterpret_cast<long *>(len_out), reinterpret_cast<const long *>(lengths), batch, quantum, add, sm);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}
