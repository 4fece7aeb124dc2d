// Host-side helpers shared by the model-level handles (see model_common.h).
#include "model_common.h"

namespace ttsamd {
namespace model {

const HostTensor *find_tensor(const TensorMap &t, const std::string &name)
{
    auto it = t.find(name);
    return it == t.end() ? nullptr : &it->second;
}

int fold_weight_norm(const TensorMap &t, const char *who, const std::string &name, HostTensor &out)
{
    if (const HostTensor *w = find_tensor(t, name + ".weight")) {
        out = *w;
        return TTSAMD_OK;
    }
    const HostTensor *g = find_tensor(t, name + ".parametrizations.weight.original0");
    const HostTensor *v = find_tensor(t, name + ".parametrizations.weight.original1");
    if (!g) g = find_tensor(t, name + ".weight_g");
    if (!v) v = find_tensor(t, name + ".weight_v");
    if (!g || !v) {
        set_error("%s: no weight for '%s' (expected .weight, .parametrizations.weight.original0/1 or .weight_g/_v)", who, name.c_str());
        return TTSAMD_ERR_INVALID;
    }
    // torch weight_norm, dim = 0: w = v * g / ||v|| with the norm over every dim but 0 (for ConvTranspose1d dim 0 is in_channels)
    out = *v;
    const int64_t d0 = out.shape[0], inner = out.numel() / d0;
    if (g->numel() != d0) {
        set_error("%s: weight-norm gain of '%s' has %lld elements, the weight has %lld rows", who, name.c_str(), (long long)g->numel(), (long long)d0);
        return TTSAMD_ERR_INVALID;
    }
    for (int64_t r = 0; r < d0; ++r) {
        // torch._weight_norm: norm in fp32 (sum of squares, sqrt), then v * (g / norm)
        float ss = 0.f;
        for (int64_t i = 0; i < inner; ++i) ss += out.data[r * inner + i] * out.data[r * inner + i];
        const float scale = g->data[r] / std::sqrt(ss);
        for (int64_t i = 0; i < inner; ++i) out.data[r * inner + i] *= scale;
    }
    return TTSAMD_OK;
}

const float *opt_bias(const TensorMap &t, const char *who, const std::string &name, int64_t n, int *rc)
{
    const HostTensor *b = find_tensor(t, name + ".bias");
    if (!b) return nullptr;
    if (b->numel() != n) {
        set_error("%s: '%s.bias' has %lld elements, expected %lld", who, name.c_str(), (long long)b->numel(), (long long)n);
        *rc = TTSAMD_ERR_INVALID;
        return nullptr;
    }
    return b->data.data();
}

int need_tensor(const TensorMap &t, const char *who, const std::string &name, int64_t n, const HostTensor **out)
{
    const HostTensor *p = find_tensor(t, name);
    if (!p) {
        set_error("%s: the state_dict has no '%s'", who, name.c_str());
        return TTSAMD_ERR_INVALID;
    }
    if (n >= 0 && p->numel() != n) {
        set_error("%s: '%s' has %lld elements, the config says %lld", who, name.c_str(), (long long)p->numel(), (long long)n);
        return TTSAMD_ERR_INVALID;
    }
    *out = p;
    return TTSAMD_OK;
}

int pack_conv(PackedConv &pc, const char *who, const float *w, const float *bias, int c_out, int c_in, int kernel, int dilation, int pad_left)
{
    pc.c_out = c_out;
    pc.c_in = c_in;
    pc.kernel = kernel;
    pc.dilation = dilation;
    pc.pad_left = pad_left < 0 ? (kernel - 1) * dilation / 2 : pad_left;
    if (!ttsamd_conv1d_supported(kernel, dilation)) {
        set_error("%s: conv kernel=%d dilation=%d is outside the HIP path's range", who, kernel, dilation);
        return TTSAMD_ERR_UNSUPPORTED;
    }
    pc.tuned = ttsamd_conv1d_tuned(kernel, dilation) != 0;
    {
        std::vector<float> img(ttsamd_conv1d_packed_floats(c_out, c_in, kernel));
        RC(ttsamd_conv1d_pack_weights(img.data(), w, c_out, c_in, kernel));
        RC(pc.w.upload(img.data(), img.size() * sizeof(float)));
    }
    {
        std::vector<unsigned char> img(ttsamd_conv1d_packed_split_bytes(c_out, c_in, kernel));
        RC(ttsamd_conv1d_pack_weights_split(img.data(), w, c_out, c_in, kernel));
        RC(pc.w_split.upload(img.data(), img.size()));
    }
    if (pc.tuned) {
        std::vector<unsigned char> img(ttsamd_conv1d_packed_h2_bytes(c_out, c_in, kernel));
        RC(ttsamd_conv1d_pack_weights_h2(img.data(), w, c_out, c_in, kernel));
        RC(pc.w_h2.upload(img.data(), img.size()));
    }
    if (c_out == c_in && (c_out == 8 || c_out == 16)) {       // the fused pair's 32-channel tile reads zero-padded images
        std::vector<float> wp((size_t)32 * 32 * kernel, 0.f);
        for (int r = 0; r < c_out; ++r)
            for (int c = 0; c < c_in; ++c)
                for (int t = 0; t < kernel; ++t) wp[((size_t)r * 32 + c) * kernel + t] = w[((size_t)r * c_in + c) * kernel + t];
        std::vector<unsigned char> a(ttsamd_conv1d_packed_split_bytes(32, 32, kernel)), b(ttsamd_conv1d_packed_h2_bytes(32, 32, kernel));
        RC(ttsamd_conv1d_pack_weights_split(a.data(), wp.data(), 32, 32, kernel));
        RC(ttsamd_conv1d_pack_weights_h2(b.data(), wp.data(), 32, 32, kernel));
        RC(pc.w_split_pad32.upload(a.data(), a.size()));
        RC(pc.w_h2_pad32.upload(b.data(), b.size()));
    }
    pc.has_bias = bias != nullptr;
    if (bias) return pc.bias.upload(bias, (size_t)c_out * sizeof(float));
    return TTSAMD_OK;
}

int pack_named_conv(const TensorMap &t, const char *who, const std::string &name, PackedConv &pc, int c_out, int c_in, int kernel, int dilation, int pad_left)
{
    HostTensor w;
    RC(fold_weight_norm(t, who, name, w));
    if (w.shape.size() != 3 || w.shape[0] != c_out || w.shape[1] != c_in || w.shape[2] != kernel) {
        set_error("%s: '%s' has shape [%lld, %lld, %lld], the config says [%d, %d, %d]", who, name.c_str(), (long long)(w.shape.size() > 0 ? w.shape[0] : -1),
                  (long long)(w.shape.size() > 1 ? w.shape[1] : -1), (long long)(w.shape.size() > 2 ? w.shape[2] : -1), c_out, c_in, kernel);
        return TTSAMD_ERR_INVALID;
    }
    int rc = TTSAMD_OK;
    const float *b = opt_bias(t, who, name, c_out, &rc);
    if (rc) return rc;
    return pack_conv(pc, who, w.data.data(), b, c_out, c_in, kernel, dilation, pad_left);
}

int upload_named(const TensorMap &t, const char *who, const std::string &name, int64_t n, DevBuf &dst)
{
    const HostTensor *p = nullptr;
    RC(need_tensor(t, who, name, n, &p));
    return dst.upload(p->data.data(), p->data.size() * sizeof(float));
}

void fill_conv_args(int precision, ttsamd_conv1d_args &a, const PackedConv &pc, const float *x, int c_x, int t_in, float *y, int c_y, int t_y, int batch)
{
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.x_bstride = (int64_t)c_x * t_in;
    a.x_rstride = t_in;
    a.c_in = pc.c_in;
    a.t_in = t_in;
    a.w_packed = static_cast<const float *>(pc.w.p);
    a.bias = pc.has_bias ? static_cast<const float *>(pc.bias.p) : nullptr;
    a.c_out = pc.c_out;
    a.kernel = pc.kernel;
    a.dilation = pc.dilation;
    a.pad_left = pc.pad_left;
    a.y = y;
    a.y_bstride = (int64_t)c_y * t_y;
    a.y_rstride = t_y;
    a.t_out = (pc.kernel % 2 == 0) ? t_in + 2 * pc.pad_left - (pc.kernel - 1) * pc.dilation : t_in;
    a.batch = batch;
    a.shuffle_t_out = t_y;
    // the precision switch of tts_amd/ops.py: conv1d
    a.w_split = (precision != 2 || !pc.tuned) ? pc.w_split.p : nullptr;
    a.w_h2 = (precision == 0 && pc.tuned) ? pc.w_h2.p : nullptr;
}

void fix_conv_mode(int precision, ttsamd_conv1d_args &a, const PackedConv &pc)
{
    const int mode = a.mode;
    const bool untuned_mode = (mode == TTSAMD_CONV_SHUFFLE && pc.kernel != 2) || (mode == TTSAMD_CONV_GATE && !((pc.kernel == 3 || pc.kernel == 5) && pc.dilation == 1)) ||
                              (mode != TTSAMD_CONV_NORMAL && mode != TTSAMD_CONV_GATE && mode != TTSAMD_CONV_SHUFFLE && pc.kernel != 1);
    a.w_split = (precision != 2 || !pc.tuned || untuned_mode) ? pc.w_split.p : nullptr;
    a.w_h2 = (precision == 0 && pc.tuned && !untuned_mode) ? pc.w_h2.p : nullptr;
}

void fill_norm_args(ttsamd_norm_args &n, const float *x, float *y, int c, int t, int batch, const float *gamma, const float *beta, float eps)
{
    memset(&n, 0, sizeof(n));
    n.x = x;
    n.x_bstride = (int64_t)c * t;
    n.x_rstride = t;
    n.c = c;
    n.t = t;
    n.batch = batch;
    n.gamma = gamma;
    n.beta = beta;
    n.eps = eps;
    n.y = y;
    n.y_bstride = (int64_t)c * t;
    n.y_rstride = t;
}

GraphEntry *GraphCache::find(const std::vector<const void *> &p, const std::vector<int64_t> &i, hipStream_t st)
{
    for (auto &e : entries)
        if (e.stream == st && e.key_ptrs == p && e.key_ints == i) return &e;
    return nullptr;
}

void GraphCache::clear()
{
    for (auto &g : entries) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    entries.clear();
}

GraphCache::~GraphCache()
{
    clear();
    if (cap_stream) (void)hipStreamDestroy(cap_stream);
}

}  // namespace model
}  // namespace ttsamd
