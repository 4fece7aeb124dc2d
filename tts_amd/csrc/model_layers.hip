// Layer drivers shared by the acoustic-model handles (see model_layers.h).
#include "model_layers.h"

namespace ttsamd {
namespace model {

int upload_norm(const TensorMap &t, const char *who, const std::string &name, int c, float eps, Norm &n)
{
    RC(upload_named(t, who, name + ".gamma", c, n.gamma));
    RC(upload_named(t, who, name + ".beta", c, n.beta));
    n.eps = eps;
    return TTSAMD_OK;
}

void pair_permute(const std::vector<float> &w, const float *bias, int64_t inner, int n, int second_offset, std::vector<float> &wp, std::vector<float> &bp)
{
    const int tiles = (n + kPairRows - 1) / kPairRows;
    wp.assign((size_t)tiles * 2 * kPairRows * inner, 0.f);
    bp.assign((size_t)tiles * 2 * kPairRows, 0.f);
    for (int a = 0; a < tiles; ++a) {
        const int lo = kPairRows * a, cnt = std::min(kPairRows, n - lo);
        for (int half = 0; half < 2; ++half)
            for (int i = 0; i < cnt; ++i) {
                const int64_t src = (half ? second_offset : 0) + lo + i, dst = (int64_t)(2 * a + half) * kPairRows + i;
                std::copy(w.begin() + src * inner, w.begin() + (src + 1) * inner, wp.begin() + dst * inner);
                if (bias) bp[dst] = bias[src];
            }
    }
}

int norm(const Ctx &c, const float *x, float *y, int ch, int t, const Norm &n, int act, const float *out_mask, const float *post_res, const float *dw_w,
         const float *dw_b, int dw_kernel, int dw_dilation, const float *in_mask)
{
    ttsamd_norm_args a;
    fill_norm_args(a, x, y, ch, t, c.B, n.gamma.f(), n.beta.f(), n.eps);
    a.act = act;
    a.out_mask = out_mask;
    if (post_res) {
        a.post_res = post_res;
        a.post_bstride = (int64_t)ch * t;
        a.post_rstride = t;
    }
    if (dw_w) {
        a.dw_w = dw_w;
        a.dw_bias = dw_b;
        a.dw_kernel = dw_kernel;
        a.dw_dilation = dw_dilation;
        a.in_mask = in_mask;
    }
    return ttsamd_channel_norm(&a, c.s);
}

int build_wn(const TensorMap &t, const char *who, const std::string &p, int hidden, int kernel, int dilation_rate, int layers, Wn &wn)
{
    wn.in_layers.clear();
    wn.rs_layers.clear();
    int dil = 1;
    for (int i = 0; i < layers; ++i) {
        const std::string si = std::to_string(i);
        HostTensor w;
        RC(fold_weight_norm(t, who, p + "in_layers." + si, w));
        if (w.shape.size() != 3 || w.shape[0] != 2 * hidden || w.shape[1] != hidden || w.shape[2] != kernel) {
            set_error("%s: '%sin_layers.%d' does not have the shape [%d, %d, %d]", who, p.c_str(), i, 2 * hidden, hidden, kernel);
            return TTSAMD_ERR_INVALID;
        }
        int rc = TTSAMD_OK;
        const float *b = opt_bias(t, who, p + "in_layers." + si, 2 * hidden, &rc);
        if (rc) return rc;
        std::vector<float> wp, bp;
        pair_permute(w.data, b, (int64_t)hidden * kernel, hidden, hidden, wp, bp);        // ops.gate_permute
        auto in = std::make_unique<PackedConv>();
        RC(pack_conv(*in, who, wp.data(), b ? bp.data() : nullptr, (int)(bp.size()), hidden, kernel, dil, -1));
        wn.in_layers.push_back(std::move(in));
        dil *= dilation_rate;
        auto rs = std::make_unique<PackedConv>();
        const int rs_out = (i < layers - 1) ? 2 * hidden : hidden;       // wavenet.py:82-85
        RC(pack_named_conv(t, who, p + "res_skip_layers." + si, *rs, rs_out, hidden, 1, 1));
        wn.rs_layers.push_back(std::move(rs));
    }
    return TTSAMD_OK;
}

int run_wn(const Ctx &c, const Wn &w, float *h, float *acts, float *out, const float *mask, int H, int t)
{
    const int n = (int)w.in_layers.size();
    for (int i = 0; i < n; ++i) {
        ttsamd_conv1d_args a;
        fill_conv_args(c.precision, a, *w.in_layers[i], h, H, t, acts, H, t, c.B);
        a.mode = TTSAMD_CONV_GATE;
        fix_conv_mode(c.precision, a, *w.in_layers[i]);
        RC(conv(c, a));
        if (i < n - 1) {
            fill_conv_args(c.precision, a, *w.rs_layers[i], acts, H, t, h, H, t, c.B);
            a.mode = TTSAMD_CONV_RES_SKIP;
            fix_conv_mode(c.precision, a, *w.rs_layers[i]);
            a.res = h;
            a.res_bstride = (int64_t)H * t;
            a.res_rstride = t;
            a.out_mask = mask;
            a.y2 = out;
            a.y2_bstride = (int64_t)H * t;
            a.y2_rstride = t;
            if (i > 0) {
                a.accum = out;
                a.accum_bstride = (int64_t)H * t;
                a.accum_rstride = t;
            }
            a.split_row = H;
        } else {
            fill_conv_args(c.precision, a, *w.rs_layers[i], acts, H, t, out, H, t, c.B);
            if (i > 0) {
                a.accum = out;
                a.accum_bstride = (int64_t)H * t;
                a.accum_rstride = t;
            }
            a.out_mask = mask;
        }
        RC(conv(c, a));
    }
    return TTSAMD_OK;
}

int build_transformer(const TensorMap &t, const char *who, const std::string &p, int H, int ffn, int heads, int layers, int kernel, int window, float eps,
                      Transformer &out)
{
    out.layers.clear();
    out.hidden = H;
    out.ffn = ffn;
    out.heads = heads;
    out.window = window;
    const int dk = H / heads;
    if (find_tensor(t, p + "proj.weight")) {
        set_error("%s: '%sproj' (a transformer whose out_channels differ from its hidden size) has no handle path", who, p.c_str());
        return TTSAMD_ERR_UNSUPPORTED;
    }
    for (int i = 0; i < layers; ++i) {
        auto L = std::make_unique<EncLayer>();
        const std::string si = std::to_string(i), a = p + "attn_layers." + si + ".", f = p + "ffn_layers." + si + ".";
        // one fused projection launch: rows q | k | v (transformer.py:106-110)
        std::vector<float> w, b;
        for (const char *n : {"conv_q", "conv_k", "conv_v"}) {
            HostTensor wt;
            RC(fold_weight_norm(t, who, a + n, wt));
            if (wt.numel() != (int64_t)H * H) {
                set_error("%s: '%s%s' is not [%d, %d, 1]", who, a.c_str(), n, H, H);
                return TTSAMD_ERR_INVALID;
            }
            w.insert(w.end(), wt.data.begin(), wt.data.end());
            const HostTensor *bt = nullptr;
            RC(need_tensor(t, who, a + n + ".bias", H, &bt));
            b.insert(b.end(), bt->data.begin(), bt->data.end());
        }
        RC(pack_conv(L->qkv, who, w.data(), b.data(), 3 * H, H, 1, 1, -1));
        RC(pack_named_conv(t, who, a + "conv_o", L->o, H, H, 1, 1));
        if (window > 0) {
            RC(upload_named(t, who, a + "emb_rel_k", (int64_t)(2 * window + 1) * dk, L->emb_k));     // heads_share: [1, 2w+1, dk]
            RC(upload_named(t, who, a + "emb_rel_v", (int64_t)(2 * window + 1) * dk, L->emb_v));
        }
        RC(upload_norm(t, who, p + "norm_layers_1." + si, H, eps, L->n1));
        RC(pack_named_conv(t, who, f + "conv_1", L->f1, ffn, H, kernel, 1, (kernel - 1) / 2));
        RC(pack_named_conv(t, who, f + "conv_2", L->f2, H, ffn, kernel, 1, (kernel - 1) / 2));
        RC(upload_norm(t, who, p + "norm_layers_2." + si, H, eps, L->n2));
        out.layers.push_back(std::move(L));
    }
    return TTSAMD_OK;
}

int run_transformer(const Ctx &cx, const Transformer &tr, const TransformerBufs &b, const float *x_mask, float **out)
{
    const int H = tr.hidden, F = tr.ffn, B = cx.B, T = cx.T;
    float *xc = b.xa, *xn = b.xb;
    ttsamd_conv1d_args a;
    for (const auto &Lp : tr.layers) {
        const EncLayer &L = *Lp;
        fill_conv_args(cx.precision, a, L.qkv, xc, H, T, b.qkv, 3 * H, T, B);
        RC(conv(cx, a));
        RC(ttsamd_rel_attention(b.att, b.qkv, b.qkv + (size_t)H * T, b.qkv + (size_t)2 * H * T, (int64_t)3 * H * T, x_mask, tr.window ? L.emb_k.f() : nullptr,
                                tr.window ? L.emb_v.f() : nullptr, tr.window, B, tr.heads, H / tr.heads, T, cx.s));
        fill_conv_args(cx.precision, a, L.o, b.att, H, T, b.xy, H, T, B);           // x + attn(x)
        a.res = xc;
        a.res_bstride = (int64_t)H * T;
        a.res_rstride = T;
        RC(conv(cx, a));
        RC(norm(cx, b.xy, b.x1, H, T, L.n1, TTSAMD_ACT_NONE, x_mask));
        fill_conv_args(cx.precision, a, L.f1, b.x1, H, T, b.hid, F, T, B);          // relu(conv_1(x * mask)) * mask
        a.out_act = TTSAMD_ACT_RELU;
        a.out_mask = x_mask;
        a.t_out = T;
        RC(conv(cx, a));
        fill_conv_args(cx.precision, a, L.f2, b.hid, F, T, b.y, H, T, B);           // x + ffn(x)
        a.res = b.x1;
        a.res_bstride = (int64_t)H * T;
        a.res_rstride = T;
        a.out_mask = x_mask;
        a.t_out = T;
        RC(conv(cx, a));
        RC(norm(cx, b.y, xn, H, T, L.n2, TTSAMD_ACT_NONE, x_mask));
        std::swap(xc, xn);
    }
    *out = xc;
    return TTSAMD_OK;
}

int build_dp(const TensorMap &t, const char *who, const std::string &p, int in_channels, int hidden, Dp &dp)
{
    dp.hidden = hidden;
    if (find_tensor(t, p + "cond.weight") || find_tensor(t, p + "cond_lang.weight")) {
        set_error("%s: a conditioned duration predictor ('%scond') has no handle path", who, p.c_str());
        return TTSAMD_ERR_UNSUPPORTED;
    }
    RC(pack_named_conv(t, who, p + "conv_1", dp.c1, hidden, in_channels, 3, 1));
    RC(pack_named_conv(t, who, p + "conv_2", dp.c2, hidden, hidden, 3, 1));
    RC(upload_norm(t, who, p + "norm_1", hidden, 1e-4f, dp.n1));
    RC(upload_norm(t, who, p + "norm_2", hidden, 1e-4f, dp.n2));
    RC(pack_named_conv(t, who, p + "proj", dp.proj, 1, hidden, 1, 1));
    return TTSAMD_OK;
}

int run_dp(const Ctx &cx, const Dp &dp, const float *x, int c_x, const float *mask, float *const d[4], float *logw)
{
    const int D = dp.hidden, B = cx.B, T = cx.T;
    ttsamd_conv1d_args a;
    fill_conv_args(cx.precision, a, dp.c1, x, c_x, T, d[0], D, T, B);
    a.in_mask = mask;
    a.out_act = TTSAMD_ACT_RELU;
    RC(conv(cx, a));
    RC(norm(cx, d[0], d[1], D, T, dp.n1));
    fill_conv_args(cx.precision, a, dp.c2, d[1], D, T, d[2], D, T, B);
    a.in_mask = mask;
    a.out_act = TTSAMD_ACT_RELU;
    RC(conv(cx, a));
    RC(norm(cx, d[2], d[3], D, T, dp.n2));
    fill_conv_args(cx.precision, a, dp.proj, d[3], D, T, logw, 1, T, B);
    a.in_mask = mask;
    a.out_mask = mask;
    return conv(cx, a);
}

}  // namespace model
}  // namespace ttsamd
