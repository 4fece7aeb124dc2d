// Shared host-side pieces of the model-level handles (hifigan_model.hip, vits_model.hip, glow_model.hip): host tensors loaded under
// the reference's state_dict key names, weight-norm folding, packing one conv layer into the three fragment images, device buffers,
// the bump workspace, and helpers that fill the kernel-level ABI structs.  Host code only: a handle issues exactly the kernel-level
// ABI calls (include/tts_amd.h) the Python host issues, so the arithmetic and the bits are those of the Python-driven path.
// Everything here may throw std::bad_alloc / length_error: every extern "C" entry of a handle runs behind abi_guard (common.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.h"

namespace ttsamd {
namespace model {

#define RC(call)               \
    do {                       \
        int rc_ = (call);      \
        if (rc_) return rc_;   \
    } while (0)

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    int64_t numel() const
    {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};
using TensorMap = std::map<std::string, HostTensor>;

// ttsamd_*_load: one state_dict entry (fp32, contiguous, host) under its reference key name
inline int load_tensor(TensorMap &tensors, const char *who, const char *name, const float *data, const int64_t *shape, int ndim)
{
    TTSAMD_CHECK_ARG(name && data && shape && ndim >= 1 && ndim <= 4, "%s_load: bad arguments", who);
    HostTensor t;
    t.shape.assign(shape, shape + ndim);
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        TTSAMD_CHECK_ARG(shape[i] > 0 && shape[i] <= (int64_t)1 << 31, "%s_load: '%s' has a dimension of %lld", who, name, (long long)shape[i]);
        n *= shape[i];
        TTSAMD_CHECK_ARG(n <= (int64_t)1 << 33, "%s_load: '%s' is larger than 2^33 elements", who, name);
    }
    t.data.assign(data, data + n);
    tensors[name] = std::move(t);
    return TTSAMD_OK;
}

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    int alloc(size_t n)
    {
        release();
        TTSAMD_HIP(hipMalloc(&p, n ? n : 4));
        bytes = n;
        return TTSAMD_OK;
    }
    int upload(const void *src, size_t n)
    {
        RC(alloc(n));
        if (n) TTSAMD_HIP(hipMemcpy(p, src, n, hipMemcpyHostToDevice));
        return TTSAMD_OK;
    }
    const float *f() const { return static_cast<const float *>(p); }
};

// one conv layer on the device: the three fragment images + bias (tts_amd/ops.py: PackedConv)
struct PackedConv {
    int c_out = 0, c_in = 0, kernel = 0, dilation = 1, pad_left = 0;
    bool tuned = false;
    DevBuf w, w_split, w_h2, bias, w_split_pad32, w_h2_pad32;
    bool has_bias = false;
};

const HostTensor *find_tensor(const TensorMap &t, const std::string &name);
// `name` + ".weight", or torch weight_norm's pair (".parametrizations.weight.original0/1" or ".weight_g/_v") folded: w = v * g / ||v||,
// norm over every dim but 0 (tts_amd/ops.py: fold_weight_norm)
int fold_weight_norm(const TensorMap &t, const char *who, const std::string &name, HostTensor &out);
// `name`.bias with n elements, or nullptr (rc set on a size mismatch)
const float *opt_bias(const TensorMap &t, const char *who, const std::string &name, int64_t n, int *rc);
// a tensor that must exist with exactly `n` elements
int need_tensor(const TensorMap &t, const char *who, const std::string &name, int64_t n, const HostTensor **out);
// w [c_out, c_in, kernel] (host) -> fp32 / split-bf16 / two-part fp16 images (+ the zero-padded 32-channel images of 8- / 16-channel
// ResBlock pairs) on the device; pad_left < 0: (kernel - 1) * dilation / 2
int pack_conv(PackedConv &pc, const char *who, const float *w, const float *bias, int c_out, int c_in, int kernel, int dilation, int pad_left);
// a plain Conv1d layer `name` of the state_dict (weight-norm folded if parametrised), shape checked against the config
int pack_named_conv(const TensorMap &t, const char *who, const std::string &name, PackedConv &pc, int c_out, int c_in, int kernel, int dilation,
                    int pad_left = -1);
// 1-D device copy of a host tensor `name` (n elements expected; n < 0: any)
int upload_named(const TensorMap &t, const char *who, const std::string &name, int64_t n, DevBuf &dst);

// conv precision of a handle: 0 = h2 (three fp16 products on large grids), 1 = x3 (six bf16 products), 2 = f32
// the precision switch and the untuned-mode rule of tts_amd/ops.py: conv1d
void fill_conv_args(int precision, ttsamd_conv1d_args &a, const PackedConv &pc, const float *x, int c_x, int t_in, float *y, int c_y, int t_y, int batch);
// call after setting a.mode: modes without a tuned instantiation for this (kernel, dilation) run on the generic split-bf16 kernel
void fix_conv_mode(int precision, ttsamd_conv1d_args &a, const PackedConv &pc);

void fill_norm_args(ttsamd_norm_args &n, const float *x, float *y, int c, int t, int batch, const float *gamma, const float *beta, float eps);

// bump allocator over a handle's grow-only device workspace; a size pass (dry) computes the bytes
struct Bump {
    unsigned char *base = nullptr;
    size_t used = 0;
    bool dry = true;
    float *take(size_t floats)
    {
        const size_t bytes = (floats * 4 + 255) & ~size_t(255);
        float *p = dry ? nullptr : reinterpret_cast<float *>(base + used);
        used += bytes;
        return p;
    }
    template <class T>
    T *take_as(size_t n)
    {
        return reinterpret_cast<T *>(take((n * sizeof(T) + 3) / 4));
    }
};

// captured launch sequences of a handle, keyed by everything a replay depends on
struct GraphEntry {
    std::vector<const void *> key_ptrs;
    std::vector<int64_t> key_ints;
    hipStream_t stream = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};
struct GraphCache {
    std::vector<GraphEntry> entries;
    hipStream_t cap_stream = nullptr;       // sequences are RECORDED here (the caller's stream may be the NULL stream) and replayed on the caller's
    size_t max_entries = 16;
    GraphEntry *find(const std::vector<const void *> &p, const std::vector<int64_t> &i, hipStream_t st);
    // run `body(stream)` once eagerly on `st` is the CALLER's job; this records it on cap_stream and stores the executable graph
    template <class F>
    int capture(const std::vector<const void *> &p, const std::vector<int64_t> &i, hipStream_t st, F &&body);
    void clear();
    ~GraphCache();
};

template <class F>
int GraphCache::capture(const std::vector<const void *> &p, const std::vector<int64_t> &i, hipStream_t st, F &&body)
{
    GraphEntry e;
    e.key_ptrs = p;
    e.key_ints = i;
    e.stream = st;
    if (!cap_stream) TTSAMD_HIP(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
    TTSAMD_HIP(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = body(cap_stream);
    const hipError_t he = hipStreamEndCapture(cap_stream, &e.graph);
    if (rc || he != hipSuccess) {
        if (e.graph) (void)hipGraphDestroy(e.graph);
        if (rc) return rc;
        TTSAMD_HIP(he);
    }
    const hipError_t hi = hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0);
    if (hi != hipSuccess) {
        (void)hipGraphDestroy(e.graph);
        TTSAMD_HIP(hi);
    }
    if (entries.size() >= max_entries) {
        GraphEntry &old = entries.front();
        // the oldest entry's stream may have been destroyed by its owner since: then wait for the whole device instead
        if (hipStreamSynchronize(old.stream) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();
        }
        (void)hipGraphExecDestroy(old.exec);
        (void)hipGraphDestroy(old.graph);
        entries.erase(entries.begin());
    }
    entries.push_back(std::move(e));
    return TTSAMD_OK;
}

}  // namespace model
}  // namespace ttsamd
