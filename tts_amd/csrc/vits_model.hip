// Model-level C ABI of VITS ("mi355_vits_{create,load,infer,destroy}"): Vits.inference of the single-speaker
// LJSpeech model behind ONE handle — weight folding / re-ordering / packing at finalize (TTS/tts/models/vits.py:653-724,1698-1725),
// then the launch sequence of vits.py:1088-1173 in two calls around the request's one host wait (include/tts_amd.h):
//
//   encode  embed -> 6 x [qkv 1x1 -> relative attention -> conv_o + x -> LayerNorm2 -> FFN conv k3 (ReLU) -> conv k3 + x -> LayerNorm2]
//           -> proj (m | logs)                                                                      networks.py:29-100, transformer.py
//           -> StochasticDurationPredictor reverse (pre, DDSConv, proj, [ConvFlow x3, ElementwiseAffine]) stochastic_duration_predictor.py:222-294
//              or DurationPredictor                                                                  glow_tts/duration_predictor.py:7-69
//           -> w = exp(logw) mask length_scale, ceil, cumsum, y_lengths -> pinned host mirror        vits.py:1140-1147
//   decode  prior expansion along the path + noise, generate_path                                   vits.py:1149-1155
//           -> 4 x [pre 1x1 -> WN (gate conv k5 + res/skip 1x1) x 4 -> post 1x1 coupling], flips folded into the weights   networks.py:103-232
//           -> waveform decoder (the vocoder handle, hifigan_model.hip) on z * y_mask                vits.py:1161
//
// Every launch goes through the kernel-level ABI of this same library with the arguments the Python host (tts_amd/vits.py,
// layers.py) passes: same kernels, same tiles, therefore the same bits (tests/test_vits_gpu.py).  Ownership: the caller owns
// tokens / noise / every output; the handle owns its packed weights, a grow-only workspace and the pinned y_lengths mirror.
#include "model_layers.h"

using namespace ttsamd;
using namespace ttsamd::model;

namespace {

constexpr const char *kWho = "vits";
constexpr int kRelWindow = 4;            // TextEncoder: rel_attn_window_size=4 (networks.py:66-76)
constexpr int kSdpHidden = 192;          // StochasticDurationPredictor(hidden + lang, 192, 3, p, 4) (vits.py:684-692)
constexpr int kSdpKernel = 3;
constexpr int kSdpFlows = 4;
constexpr int kGraphTailMaxFrames = 2048;    // B * padded frames up to which a single request's tail is captured (tts_amd/vits.py)
constexpr int kDpHidden = 256;           // DurationPredictor(hidden + lang, 256, 3, p) (vits.py:694-702)

struct DdsLayer {
    DevBuf dw_w, dw_b;
    int dil = 1;
    PackedConv pw;
    Norm n1, n2;
};
using Dds = std::vector<std::unique_ptr<DdsLayer>>;

struct SdpFlow {
    DevBuf pre_w, pre_b;
    Dds convs;
    PackedConv proj;
};

struct Flow {
    bool flipped = false;
    PackedConv pre, post;
    Wn wn;
};

struct Model {
    ttsamd_vits_config cfg{};
    TensorMap tensors;
    bool finalized = false;
    // text encoder
    DevBuf emb;
    Transformer enc;
    PackedConv te_proj;
    // duration predictor
    PackedConv dp_pre, dp_proj;           // SDP: pre / proj; DP: unused / proj
    Dds dp_convs;
    DevBuf ea_m, ea_logs;
    std::vector<std::unique_ptr<SdpFlow>> sdp_flows;     // index i - 1 holds flows.i
    int num_bins = 10;
    Dp dp;                                // DurationPredictor (use_sdp = 0)
    // flow
    std::vector<std::unique_ptr<Flow>> flows;
    // waveform decoder: a child vocoder handle
    void *decoder = nullptr;
    int64_t hop = 1;
    // request state (encode -> decode)
    DevBuf work;                           // encode-phase activations + the tensors decode reads
    DevBuf work2;                          // decode-phase activations
    DevBuf work3;                          // static buffers of the captured single-request tail
    int64_t *host_len = nullptr;           // pinned mirror of y_lengths
    int host_len_cap = 0;
    struct Req {
        bool valid = false;
        int B = 0, T = 0, t_dec = 0;
        const float *x_mask = nullptr, *stats = nullptr, *h = nullptr, *logw = nullptr;
        float *w_ceil = nullptr;
        int32_t *cum = nullptr;
        int64_t *y_lengths = nullptr;
    } req;
    GraphCache front_graphs, tail_graphs;
    ~Model()
    {
        if (decoder) (void)ttsamd_hifigan_destroy(decoder);
        if (host_len) (void)hipHostFree(host_len);
    }
};

Model *as_model(void *h) { return static_cast<Model *>(h); }

int build_dds(const Model &m, const std::string &p, int channels, int kernel, int layers, Dds &out)
{
    out.clear();
    int dil = 1;
    for (int i = 0; i < layers; ++i) {
        auto L = std::make_unique<DdsLayer>();
        const std::string si = std::to_string(i);
        RC(upload_named(m.tensors, kWho, p + "convs_sep." + si + ".weight", (int64_t)channels * kernel, L->dw_w));
        RC(upload_named(m.tensors, kWho, p + "convs_sep." + si + ".bias", channels, L->dw_b));
        L->dil = dil;                                    // kernel_size ** i (stochastic_duration_predictor.py:36)
        dil *= kernel;
        RC(pack_named_conv(m.tensors, kWho, p + "convs_1x1." + si, L->pw, channels, channels, 1, 1));
        RC(upload_norm(m.tensors, kWho, p + "norms_1." + si, channels, 1e-5f, L->n1));
        RC(upload_norm(m.tensors, kWho, p + "norms_2." + si, channels, 1e-5f, L->n2));
        out.push_back(std::move(L));
    }
    return TTSAMD_OK;
}

int finalize(Model &m)
{
    const ttsamd_vits_config &c = m.cfg;
    const int H = c.hidden_channels, heads = c.num_heads_text_encoder, kte = c.kernel_size_text_encoder;
    m.flows.clear();
    m.sdp_flows.clear();
    m.front_graphs.clear();
    m.tail_graphs.clear();
    m.req.valid = false;
    RC(upload_named(m.tensors, kWho, "text_encoder.emb.weight", (int64_t)c.num_chars * H, m.emb));
    RC(build_transformer(m.tensors, kWho, "text_encoder.encoder.", H, c.hidden_channels_ffn_text_encoder, heads, c.num_layers_text_encoder, kte, kRelWindow,
                         1e-5f, m.enc));        // LayerNorm2 (eps 1e-5), rel_attn_window_size 4
    RC(pack_named_conv(m.tensors, kWho, "text_encoder.proj", m.te_proj, 2 * H, H, 1, 1));
    const std::string dp = "duration_predictor.";
    if (c.use_sdp) {
        RC(pack_named_conv(m.tensors, kWho, dp + "pre", m.dp_pre, kSdpHidden, H, 1, 1));
        RC(build_dds(m, dp + "convs.", kSdpHidden, kSdpKernel, 3, m.dp_convs));
        RC(pack_named_conv(m.tensors, kWho, dp + "proj", m.dp_proj, kSdpHidden, kSdpHidden, 1, 1));
        RC(upload_named(m.tensors, kWho, dp + "flows.0.translation", 2, m.ea_m));
        RC(upload_named(m.tensors, kWho, dp + "flows.0.log_scale", 2, m.ea_logs));
        for (int i = 1; i <= kSdpFlows; ++i) {
            auto F = std::make_unique<SdpFlow>();
            const std::string q = dp + "flows." + std::to_string(i) + ".";
            RC(upload_named(m.tensors, kWho, q + "pre.weight", kSdpHidden, F->pre_w));
            RC(upload_named(m.tensors, kWho, q + "pre.bias", kSdpHidden, F->pre_b));
            RC(build_dds(m, q + "convs.", kSdpHidden, kSdpKernel, 3, F->convs));
            const HostTensor *pw = nullptr;
            RC(need_tensor(m.tensors, kWho, q + "proj.weight", -1, &pw));
            const int rows = (int)pw->shape[0];
            m.num_bins = (rows + 1) / 3;
            RC(pack_named_conv(m.tensors, kWho, q + "proj", F->proj, rows, kSdpHidden, 1, 1));
            m.sdp_flows.push_back(std::move(F));
        }
    } else {
        RC(build_dp(m.tensors, kWho, dp, H, kDpHidden, m.dp));
    }
    // ResidualCouplingBlocks: the channel flip before every flow (networks.py:229-231) is folded into the weights — a flow that sees
    // a flipped tensor reads its conditioning half through input-channel-reversed `pre` weights and writes the coupled half through
    // output-row-reversed `post` weights (tts_amd/layers.py: ResidualCouplingBlocks)
    const int half = H / 2;
    for (int i = 0; i < c.num_flows; ++i) {
        auto F = std::make_unique<Flow>();
        const std::string q = "flow.flows." + std::to_string(i) + ".";
        F->flipped = (c.num_flows - i) % 2 == 1;
        const HostTensor *wpre = nullptr, *bpre = nullptr, *wpost = nullptr, *bpost = nullptr;
        RC(need_tensor(m.tensors, kWho, q + "pre.weight", (int64_t)H * half, &wpre));
        RC(need_tensor(m.tensors, kWho, q + "pre.bias", H, &bpre));
        RC(need_tensor(m.tensors, kWho, q + "post.weight", -1, &wpost));
        if (wpost->numel() != (int64_t)half * H) {
            set_error("vits: '%spost.weight' has %lld elements: only mean_only=True coupling (the VITS default) has a HIP path", q.c_str(), (long long)wpost->numel());
            return TTSAMD_ERR_UNSUPPORTED;
        }
        RC(need_tensor(m.tensors, kWho, q + "post.bias", half, &bpost));
        std::vector<float> w1 = wpre->data, w2 = wpost->data, b2 = bpost->data;
        if (F->flipped) {
            for (int r = 0; r < H; ++r) std::reverse(w1.begin() + (size_t)r * half, w1.begin() + (size_t)(r + 1) * half);       // flip(wpre, [1])
            for (int r = 0; r < half / 2; ++r) {                                                                               // flip(wpost, [0])
                std::swap_ranges(w2.begin() + (size_t)r * H, w2.begin() + (size_t)(r + 1) * H, w2.begin() + (size_t)(half - 1 - r) * H);
                std::swap(b2[r], b2[half - 1 - r]);
            }
        }
        RC(pack_conv(F->pre, kWho, w1.data(), bpre->data.data(), H, half, 1, 1, -1));
        RC(pack_conv(F->post, kWho, w2.data(), b2.data(), half, H, 1, 1, -1));
        RC(build_wn(m.tensors, kWho, q + "enc.", H, c.kernel_size_flow, c.dilation_rate_flow, c.num_layers_flow, F->wn));
        m.flows.push_back(std::move(F));
    }
    // waveform decoder: a vocoder handle fed with the "waveform_decoder." entries
    if (m.decoder) {
        (void)ttsamd_hifigan_destroy(m.decoder);
        m.decoder = nullptr;
    }
    RC(ttsamd_hifigan_create(&c.decoder, &m.decoder));
    const std::string pre = "waveform_decoder.";
    for (const auto &kv : m.tensors)
        if (kv.first.compare(0, pre.size(), pre) == 0)
            RC(ttsamd_hifigan_load(m.decoder, kv.first.c_str() + pre.size(), kv.second.data.data(), kv.second.shape.data(), (int)kv.second.shape.size()));
    RC(ttsamd_hifigan_finalize(m.decoder));
    m.hop = 1;
    for (int i = 0; i < c.decoder.num_upsamples; ++i) m.hop *= c.decoder.upsample_factors[i];
    m.tensors.clear();
    m.finalized = true;
    return TTSAMD_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// launch sequences (tts_amd/layers.py)
// ---------------------------------------------------------------------------------------------------------------------------
// DilatedDepthSeparableConv (stochastic_duration_predictor.py:46-63): x [B,C,T] (conditioning already added) -> DDSConv(x) * mask.
// bufs: four [B,C,T] buffers; the input lives in bufs[0]; returns the buffer holding the result
int dds(const Ctx &c, const Dds &layers, float *const bufs[4], const float *mask, int ch, float **out)
{
    float *x = bufs[0], *alt = bufs[1], *t1 = bufs[2], *t2 = bufs[3];
    const int n = (int)layers.size();
    for (int i = 0; i < n; ++i) {
        const DdsLayer &L = *layers[i];
        RC(norm(c, x, t1, ch, c.T, L.n1, TTSAMD_ACT_GELU, nullptr, nullptr, L.dw_w.f(), L.dw_b.f(), kSdpKernel, L.dil, mask));
        ttsamd_conv1d_args a;
        fill_conv_args(c.precision, a, L.pw, t1, ch, c.T, t2, ch, c.T, c.B);
        RC(conv(c, a));
        RC(norm(c, t2, alt, ch, c.T, L.n2, TTSAMD_ACT_GELU, i == n - 1 ? mask : nullptr, x));
        std::swap(x, alt);
    }
    *out = x;
    return TTSAMD_OK;
}

// text encoder + duration predictor (tts_amd/vits.py: _front_eager).  ws.dry: only the workspace size is computed; otherwise the
// request's tensors are laid out in the workspace (m.req) and, with `launch`, the kernels are issued on `st` (a graph replay needs
// the layout only)
int front(Model &m, Bump &ws, const int64_t *x, const int64_t *x_lengths, const float *noise_dp, bool run_dp, hipStream_t st, bool launch)
{
    const ttsamd_vits_config &c = m.cfg;
    const int B = m.req.B, T = m.req.T, H = c.hidden_channels, F = c.hidden_channels_ffn_text_encoder;
    Ctx cx{c.decoder.precision, reinterpret_cast<void *>(st), B, T};
    const size_t n = (size_t)B * T;
    float *x_mask = ws.take(n);
    TransformerBufs tb;
    tb.take(ws, n, H, F);
    float *stats = ws.take(n * 2 * H);
    float *logw = ws.take(n);
    // duration-predictor buffers
    const int D = c.use_sdp ? kSdpHidden : kDpHidden;
    float *d0 = ws.take(n * D), *d1 = ws.take(n * D), *d2 = ws.take(n * D), *d3 = ws.take(n * D), *cond = ws.take(n * D);
    float *e0 = ws.take(n * D), *e1 = ws.take(n * D), *e2 = ws.take(n * D), *e3 = ws.take(n * D);
    float *par = ws.take(n * (3 * m.num_bins - 1)), *za = ws.take(n * 2), *zb = ws.take(n * 2);
    float *w_ceil = ws.take(n);
    int32_t *cum = ws.take_as<int32_t>(n);
    int64_t *ylen = ws.take_as<int64_t>(B);
    if (ws.dry) return TTSAMD_OK;
    m.req.x_mask = x_mask;
    m.req.stats = stats;
    m.req.h = transformer_result(m.enc, tb);           // the layers ping-pong between two buffers
    m.req.logw = run_dp ? logw : nullptr;
    m.req.w_ceil = w_ceil;
    m.req.cum = cum;
    m.req.y_lengths = ylen;
    if (!launch) return TTSAMD_OK;
    void *s = cx.s;
    RC(ttsamd_sequence_mask(x_mask, x_lengths, B, T, s));
    RC(ttsamd_embed(tb.xa, x, m.emb.f(), x_mask, (float)std::sqrt((double)H), B, H, T, c.num_chars, s));       // emb(x) * sqrt(H) (networks.py:87), masked
    float *xc = nullptr;
    RC(run_transformer(cx, m.enc, tb, x_mask, &xc));
    ttsamd_conv1d_args a;
    fill_conv_args(cx.precision, a, m.te_proj, xc, H, T, stats, 2 * H, T, B);
    a.out_mask = x_mask;
    RC(conv(cx, a));
    if (!run_dp) return TTSAMD_OK;
    if (c.use_sdp) {
        // StochasticDurationPredictor.forward(reverse=True), stochastic_duration_predictor.py:222-294
        fill_conv_args(cx.precision, a, m.dp_pre, xc, H, T, d0, D, T, B);
        RC(conv(cx, a));
        float *const bufs[4] = {d0, d1, d2, d3};
        float *hd = nullptr;
        RC(dds(cx, m.dp_convs, bufs, x_mask, D, &hd));
        fill_conv_args(cx.precision, a, m.dp_proj, hd, D, T, cond, D, T, B);
        a.out_mask = x_mask;
        RC(conv(cx, a));
        const float *z = noise_dp;
        if (c.inference_noise_scale_dp != 1.0f) {
            RC(ttsamd_scale(za, noise_dp, c.inference_noise_scale_dp, (int64_t)n * 2, s));
            z = za;
        }
        // flows reversed, the "useless" one dropped (:285-286): [4, 3, 2, 0]
        float *zo = (z == za) ? zb : za;
        for (int i : {kSdpFlows, kSdpFlows - 1, kSdpFlows - 2, 0}) {
            if (i == 0) {
                RC(ttsamd_sdp_affine_reverse(zo, z, m.ea_m.f(), m.ea_logs.f(), x_mask, B, T, s));
            } else {
                const SdpFlow &Fl = *m.sdp_flows[i - 1];
                RC(ttsamd_convflow_pre(e0, z, 1, Fl.pre_w.f(), Fl.pre_b.f(), cond, B, D, T, s));       // pre(x0) + g, x0 = flip(z)[:, 0] = z[:, 1]
                float *const eb[4] = {e0, e1, e2, e3};
                float *he = nullptr;
                RC(dds(cx, Fl.convs, eb, x_mask, D, &he));
                fill_conv_args(cx.precision, a, Fl.proj, he, D, T, par, 3 * m.num_bins - 1, T, B);
                a.out_mask = x_mask;
                RC(conv(cx, a));
                RC(ttsamd_convflow_spline_reverse(zo, z, par, x_mask, B, T, m.num_bins, (float)kSdpHidden, 5.0f, s));
            }
            z = zo;
            zo = (zo == za) ? zb : za;
        }
        // logw = z[:, 0] as a contiguous [B, T]
        ttsamd_copy_seg seg;
        memset(&seg, 0, sizeof(seg));
        seg.src = z;
        seg.dst = logw;
        seg.d0 = 1;
        seg.d1 = B;
        seg.d2 = T;
        seg.s0 = 0;
        seg.s1 = 2 * (int64_t)T;
        seg.s2 = 1;
        seg.t0 = 0;
        seg.t1 = T;
        seg.t2 = 1;
        seg.elem_bytes = 4;
        RC(ttsamd_copy_strided(&seg, 1, s));
    } else {
        float *const db[4] = {d0, d1, d2, d3};
        RC(model::run_dp(cx, m.dp, xc, H, x_mask, db, logw));
    }
    return TTSAMD_OK;
}

int grow(DevBuf &buf, size_t bytes)
{
    if (bytes <= buf.bytes && buf.p) return TTSAMD_OK;
    TTSAMD_HIP(hipDeviceSynchronize());
    return buf.alloc(bytes);
}

}  // namespace

extern "C" int ttsamd_vits_create(const ttsamd_vits_config *cfg, void **handle_out)
{
    return abi_guard("vits_create", [&]() -> int {
        TTSAMD_CHECK_ARG(cfg && handle_out, "vits_create: NULL argument");
        const ttsamd_vits_config &c = *cfg;
        TTSAMD_CHECK_ARG(c.num_chars > 0 && c.hidden_channels > 0 && c.hidden_channels % 2 == 0, "vits_create: bad num_chars / hidden_channels");
        TTSAMD_CHECK_ARG(c.num_heads_text_encoder > 0 && c.hidden_channels % c.num_heads_text_encoder == 0 && c.hidden_channels / c.num_heads_text_encoder <= 128,
                         "vits_create: hidden_channels %d over %d heads (head size <= 128)", c.hidden_channels, c.num_heads_text_encoder);
        TTSAMD_CHECK_ARG(c.num_layers_text_encoder >= 1 && c.num_layers_text_encoder <= 64 && c.hidden_channels_ffn_text_encoder > 0 &&
                             c.kernel_size_text_encoder >= 1 && c.kernel_size_text_encoder <= 31,
                         "vits_create: bad text-encoder configuration");
        TTSAMD_CHECK_ARG(c.num_flows >= 2 && c.num_flows <= 16 && c.num_flows % 2 == 0, "vits_create: the folded channel flips need an even number of flows (got %d)", c.num_flows);
        TTSAMD_CHECK_ARG(c.num_layers_flow >= 1 && c.num_layers_flow <= 32 && c.kernel_size_flow >= 1 && c.kernel_size_flow % 2 == 1 && c.dilation_rate_flow >= 1,
                         "vits_create: bad flow configuration");
        TTSAMD_CHECK_ARG(c.hidden_channels % kPairRows == 0, "vits_create: the gate conv's paired rows need hidden_channels %% %d == 0", kPairRows);
        TTSAMD_CHECK_ARG(c.decoder.in_channels == c.hidden_channels && c.decoder.inference_padding == 0 && c.decoder.out_channels == 1,
                         "vits_create: the waveform decoder takes hidden_channels inputs, one output channel, inference_padding 0 (vits.py:704-718)");
        TTSAMD_CHECK_ARG(c.length_scale > 0.f, "vits_create: length_scale must be positive");
        void *probe = nullptr;                 // the vocoder's own configuration checks
        RC(ttsamd_hifigan_create(&c.decoder, &probe));
        (void)ttsamd_hifigan_destroy(probe);
        Model *m = new Model();
        m->cfg = c;
        *handle_out = m;
        return TTSAMD_OK;
    });
}

extern "C" int ttsamd_vits_load(void *handle, const char *name, const float *data, const int64_t *shape, int ndim)
{
    return abi_guard("vits_load", [&]() -> int {
        TTSAMD_CHECK_ARG(handle && name, "vits_load: NULL argument");
        // discriminator: training only; posterior encoder: training / voice conversion (the Python host's path)
        if (strncmp(name, "disc.", 5) == 0 || strncmp(name, "posterior_encoder.", 18) == 0) return TTSAMD_OK;
        Model &m = *as_model(handle);
        RC(load_tensor(m.tensors, kWho, name, data, shape, ndim));
        m.finalized = false;
        return TTSAMD_OK;
    });
}

extern "C" int ttsamd_vits_finalize(void *handle)
{
    return abi_guard("vits_finalize", [&]() -> int {
        TTSAMD_CHECK_ARG(handle, "vits_finalize: NULL handle");
        Model &m = *as_model(handle);
        if (m.finalized && m.tensors.empty()) return TTSAMD_OK;
        m.finalized = false;
        TTSAMD_HIP(hipDeviceSynchronize());       // graphs / launches in flight may still read the previous weight set
        return finalize(m);
    });
}

extern "C" int64_t ttsamd_vits_hop_length(void *handle) { return handle ? as_model(handle)->hop : -1; }

extern "C" int ttsamd_vits_set_option(void *handle, int option, int value)
{
    return abi_guard("vits_set_option", [&]() -> int {
        TTSAMD_CHECK_ARG(handle && as_model(handle)->decoder, "vits_set_option: no model (ttsamd_vits_finalize first)");
        Model &m = *as_model(handle);
        if (option == TTSAMD_HIFIGAN_OPT_CONCURRENT_BRANCHES) {      // a captured tail has the branch topology baked in
            TTSAMD_HIP(hipDeviceSynchronize());
            m.tail_graphs.clear();
        }
        return ttsamd_hifigan_set_option(m.decoder, option, value);
    });
}

extern "C" int ttsamd_vits_encode(void *handle, const int64_t *x, const int64_t *x_lengths, int batch, int t_text, const float *noise_dp,
                                  const float *durations_in, int run_duration_predictor, int64_t *y_lengths_host, int32_t *t_dec_out, int use_graph,
                                  void *stream)
{
    return abi_guard("vits_encode", [&]() -> int {
        TTSAMD_CHECK_ARG(handle && x && x_lengths && t_dec_out, "vits_encode: NULL argument");
        Model &m = *as_model(handle);
        TTSAMD_CHECK_ARG(m.finalized, "vits_encode: weights not loaded (ttsamd_vits_load ... ttsamd_vits_finalize)");
        TTSAMD_CHECK_ARG(batch >= 1 && batch <= 65535 && t_text >= 1, "vits_encode: bad shape [%d, %d]", batch, t_text);
        const bool run_dp = !durations_in || run_duration_predictor;
        TTSAMD_CHECK_ARG(!(run_dp && m.cfg.use_sdp) || noise_dp, "vits_encode: the stochastic duration predictor needs noise_dp [batch, 2, t_text]");
        hipStream_t st = as_stream(stream);
        m.req = Model::Req();
        m.req.B = batch;
        m.req.T = t_text;
        Bump dry;
        RC(front(m, dry, x, x_lengths, noise_dp, run_dp, st, false));
        if (dry.used > m.work.bytes) {
            m.front_graphs.clear();               // captured sequences hold pointers into the workspace
            m.tail_graphs.clear();
            RC(grow(m.work, dry.used));
        }
        if (batch > m.host_len_cap) {
            if (m.host_len) (void)hipHostFree(m.host_len);
            m.host_len = nullptr;
            m.host_len_cap = 0;
            TTSAMD_HIP(hipHostMalloc(reinterpret_cast<void **>(&m.host_len), sizeof(int64_t) * (size_t)batch, hipHostMallocDefault));
            m.host_len_cap = batch;
        }
        auto run = [&](hipStream_t s2, bool launch) -> int {
            Bump ws;
            ws.base = static_cast<unsigned char *>(m.work.p);
            ws.dry = false;
            return front(m, ws, x, x_lengths, noise_dp, run_dp, s2, launch);
        };
        const std::vector<const void *> kp = {x, x_lengths, noise_dp};
        const std::vector<int64_t> ki = {batch, t_text, run_dp ? 1 : 0};
        GraphEntry *g = use_graph ? m.front_graphs.find(kp, ki, st) : nullptr;
        if (g) {
            RC(run(st, false));                   // lay the request out (same workspace, same shape: the addresses the graph was captured with)
            TTSAMD_HIP(hipGraphLaunch(g->exec, st));
        } else {
            RC(run(st, true));
            // first sighting of this (buffers, shape, stream): the eager run above produced this call's result; capture the same
            // sequence and replay it from the next call on
            if (use_graph) RC(m.front_graphs.capture(kp, ki, st, [&](hipStream_t s2) { return run(s2, true); }));
        }
        // durations: w = exp(logw) * mask * length_scale, ceil, cumsum, y_lengths (vits.py:1140-1147) — or the injected ones (:1141-1143);
        // y_lengths also land in the pinned mirror by a system-scope store: the host polls it instead of a reduce + D2H + stream sync
        for (int i = 0; i < batch; ++i) m.host_len[i] = -1;
        RC(ttsamd_durations_ex(m.req.w_ceil, m.req.cum, m.req.y_lengths, m.host_len, durations_in ? nullptr : m.req.logw, durations_in, m.req.x_mask,
                               durations_in ? 1.0f : m.cfg.length_scale, 0, t_text, batch, t_text, stream));
        volatile int64_t *hl = m.host_len;
        int64_t tmax = 0;
        for (int i = 0; i < batch; ++i) {
            unsigned long long spins = 0;
            while (hl[i] < 0) {
                if ((++spins & 0xFFFFF) == 0 && hipStreamQuery(st) == hipSuccess && hl[i] < 0) {
                    set_error("vits_encode: the durations kernel finished without publishing y_lengths");
                    return TTSAMD_ERR_HIP;
                }
            }
            const int64_t v = hl[i];
            tmax = std::max<int64_t>(tmax, v);
            if (y_lengths_host) y_lengths_host[i] = v;
        }
        m.req.t_dec = (int)tmax;
        m.req.valid = true;
        *t_dec_out = (int32_t)tmax;
        return TTSAMD_OK;
    });
}

namespace {

struct TailBufs {
    float *z_p, *z, *m_p, *logs_p, *y_mask, *attn, *wav, *h, *acts, *skip;
};

// Everything after the output extent is known (vits.py:1149-1161; tts_amd/vits.py: _tail_eager) at `t` frames into `b`.  ragged: the
// single-request form a graph is captured in — `t` is the 32-frame bucket of the request, the noise draw sits packed
// [B, C, max(y_lengths)] at the head of `noise`, and the decoder runs ragged-exact (every conv treats the row as ending at its own
// length), which reproduces the unpadded run.
int tail(Model &m, const TailBufs &b, const float *noise, int t, bool ragged, void *stream)
{
    const ttsamd_vits_config &c = m.cfg;
    const int B = m.req.B, T = m.req.T, H = c.hidden_channels, half = H / 2;
    Ctx cx{c.decoder.precision, stream, B, T};
    // m_p / logs_p gathered along the path, z_p = m_p + noise * exp(logs_p) * noise_scale (vits.py:1152-1155); a second copy of z_p
    // is what the flows transform in place
    RC(ttsamd_expand_prior_ex(b.z_p, b.z, b.m_p, b.logs_p, b.y_mask, m.req.stats, m.req.stats + (size_t)H * T, (int64_t)2 * H * T, noise, m.req.cum, m.req.x_mask,
                              m.req.y_lengths, c.inference_noise_scale, 0, ragged ? 1 : 0, B, H, T, t, stream));
    if (b.attn) RC(ttsamd_generate_path(b.attn, m.req.cum, m.req.x_mask, m.req.y_lengths, B, T, t, stream));
    // ResidualCouplingBlocks.forward(reverse=True), networks.py:226-231, in place on z
    ttsamd_conv1d_args a;
    for (int i = c.num_flows - 1; i >= 0; --i) {
        const Flow &F = *m.flows[i];
        const int src = F.flipped ? half : 0, dst = F.flipped ? 0 : half;
        fill_conv_args(cx.precision, a, F.pre, b.z + (size_t)src * t, H, t, b.h, H, t, B);
        a.out_mask = b.y_mask;
        RC(conv(cx, a));
        RC(run_wn(cx, F.wn, b.h, b.acts, b.skip, b.y_mask, H, t));
        // x1 = (x1 - post(h) * mask) * mask (mean_only: exp(-log_scale) == 1), in place on the other half
        fill_conv_args(cx.precision, a, F.post, b.skip, H, t, b.z + (size_t)dst * t, H, t, B);
        a.mode = TTSAMD_CONV_COUPLE;
        fix_conv_mode(cx.precision, a, F.post);
        a.res = b.z + (size_t)dst * t;
        a.res_bstride = (int64_t)H * t;
        a.res_rstride = t;
        a.out_mask = b.y_mask;
        RC(conv(cx, a));
    }
    // waveform decoder on z * y_mask (vits.py:1161): the mask rides in conv_pre's load (ragged: the decoder's own stage-0 length mask
    // IS y_mask)
    return ttsamd_hifigan_forward_ex(m.decoder, b.z, B, t, ragged ? m.req.y_lengths : nullptr, ragged ? nullptr : b.y_mask, b.wav, 0, stream);
}

void seg_box(ttsamd_copy_seg &g, void *dst, const void *src, int d0, int d1, int d2, int64_t s0, int64_t s1, int64_t t0, int64_t t1, int bytes)
{
    memset(&g, 0, sizeof(g));
    g.src = src;
    g.dst = dst;
    g.d0 = d0;
    g.d1 = d1;
    g.d2 = d2;
    g.s0 = s0;
    g.s1 = s1;
    g.s2 = 1;
    g.t0 = t0;
    g.t1 = t1;
    g.t2 = 1;
    g.elem_bytes = bytes;
}

}  // namespace

extern "C" int ttsamd_vits_decode(void *handle, const float *noise_z, const ttsamd_vits_outputs *outp, int use_graph, void *stream)
{
    return abi_guard("vits_decode", [&]() -> int {
        TTSAMD_CHECK_ARG(handle && noise_z && outp && outp->wav, "vits_decode: NULL argument (noise_z, out, out->wav)");
        Model &m = *as_model(handle);
        TTSAMD_CHECK_ARG(m.finalized && m.req.valid, "vits_decode: no request in flight (ttsamd_vits_encode first)");
        const ttsamd_vits_config &c = m.cfg;
        const ttsamd_vits_outputs &o = *outp;
        const int B = m.req.B, T = m.req.T, td = m.req.t_dec, H = c.hidden_channels;
        hipStream_t st = as_stream(stream);
        TTSAMD_CHECK_ARG((int64_t)B * H * std::max(T, td) < ((int64_t)1 << 31) && (int64_t)td * m.hop < ((int64_t)1 << 31), "vits_decode: tensors beyond 2^31 elements");
        ttsamd_copy_seg segs[TTSAMD_COPY_MAX_SEGS];
        int ns = 0;
        const int t_pad = (td + 31) / 32 * 32;
        const bool replay = use_graph && B == 1 && (int64_t)B * t_pad <= kGraphTailMaxFrames;
        const int To = o.t_text_out > 0 ? o.t_text_out : T;          // token extent of the token-indexed outputs
        TTSAMD_CHECK_ARG(To <= T && (To == T || replay), "vits_decode: t_text_out %d (request ran at %d tokens) needs a replayed tail", o.t_text_out, T);
        if (replay) {
            // A single request is launch-bound end to end: the tail (~110 launches) replays as ONE hipGraph per 32-frame bucket over
            // static buffers; the request's own launches are the noise copy in, the replay, and one copy out cut to the true extent
            // (tts_amd/vits.py: the `_tail` graph).  Outputs equal the eager run at the true length as long as the padding does not
            // move a launch across the small-grid threshold of ttsamd_conv1d_set_small_grid (fp32 summation order).
            const size_t np = (size_t)B * t_pad;
            float *noise_pad;
            TailBufs b;
            auto layout = [&](Bump &w) {
                noise_pad = w.take(np * H);
                b.z_p = w.take(np * H), b.z = w.take(np * H), b.m_p = w.take(np * H), b.logs_p = w.take(np * H), b.y_mask = w.take(np);
                b.attn = w.take((size_t)B * T * t_pad), b.wav = w.take(np * m.hop), b.h = w.take(np * H), b.acts = w.take(np * H), b.skip = w.take(np * H);
            };
            Bump ws;
            layout(ws);
            if (ws.used > m.work3.bytes) {
                m.tail_graphs.clear();            // captured sequences hold pointers into the workspace
                RC(grow(m.work3, ws.used));
            }
            ws = Bump();
            ws.base = static_cast<unsigned char *>(m.work3.p);
            ws.dry = false;
            layout(ws);
            // the draw at the reference's shape [B, C, t_dec] lands packed at the head of the bucket's noise buffer (columns beyond are
            // read as zero by the kernel: masked there anyway)
            seg_box(segs[0], noise_pad, noise_z, 1, 1, (int)((size_t)B * H * td), 0, 0, 0, 0, 4);
            RC(ttsamd_copy_strided(segs, 1, stream));
            // (the graph reads the encode phase's tensors in place: same (B, T) -> same addresses in the encode workspace)
            const std::vector<const void *> kp = {m.work.p, m.work3.p};
            const std::vector<int64_t> ki = {B, T, t_pad};
            if (GraphEntry *g = m.tail_graphs.find(kp, ki, st)) {
                TTSAMD_HIP(hipGraphLaunch(g->exec, st));
            } else {
                RC(tail(m, b, noise_pad, t_pad, true, stream));
                RC(m.tail_graphs.capture(kp, ki, st, [&](hipStream_t s2) { return tail(m, b, noise_pad, t_pad, true, reinterpret_cast<void *>(s2)); }));
            }
            // hand out copies cut to the true extent: ONE launch
            const int64_t hop = m.hop;
            seg_box(segs[ns++], o.wav, b.wav, 1, B, (int)(td * hop), 0, t_pad * hop, 0, td * hop, 4);
            if (o.alignments) seg_box(segs[ns++], o.alignments, b.attn, B, To, td, (int64_t)T * t_pad, t_pad, (int64_t)To * td, td, 4);
            const float *src4[4] = {b.z, b.z_p, b.m_p, b.logs_p};
            float *dst4[4] = {o.z, o.z_p, o.m_p, o.logs_p};
            for (int i = 0; i < 4; ++i)
                if (dst4[i]) seg_box(segs[ns++], dst4[i], src4[i], B, H, td, (int64_t)H * t_pad, t_pad, (int64_t)H * td, td, 4);
            if (o.y_mask) seg_box(segs[ns++], o.y_mask, b.y_mask, 1, B, td, 0, t_pad, 0, td, 4);
        } else {
            const size_t nt = (size_t)B * td;
            // decode-phase workspace: outputs the caller did not ask for still have to exist
            TailBufs b;
            auto layout = [&](Bump &w) {
                b.z_p = o.z_p ? o.z_p : w.take(nt * H);
                b.z = o.z ? o.z : w.take(nt * H);
                b.m_p = o.m_p ? o.m_p : w.take(nt * H);
                b.logs_p = o.logs_p ? o.logs_p : w.take(nt * H);
                b.y_mask = o.y_mask ? o.y_mask : w.take(nt);
                b.h = w.take(nt * H), b.acts = w.take(nt * H), b.skip = w.take(nt * H);
                b.attn = o.alignments;
                b.wav = o.wav;
            };
            Bump ws;
            layout(ws);
            RC(grow(m.work2, ws.used));
            ws = Bump();
            ws.base = static_cast<unsigned char *>(m.work2.p);
            ws.dry = false;
            layout(ws);
            RC(tail(m, b, noise_z, td, false, stream));
        }
        // the remaining outputs: copies of request state (same launch as the cut copies of a replayed tail)
        if (o.durations) seg_box(segs[ns++], o.durations, m.req.w_ceil, 1, B, To, 0, T, 0, To, 4);
        if (o.y_lengths) seg_box(segs[ns++], o.y_lengths, m.req.y_lengths, 1, 1, B, 0, 0, 0, 0, 8);
        if (o.logw && m.req.logw) seg_box(segs[ns++], o.logw, m.req.logw, 1, B, To, 0, T, 0, To, 4);
        if (o.x_hidden) seg_box(segs[ns++], o.x_hidden, m.req.h, 1, B * H, To, 0, T, 0, To, 4);
        if (ns) RC(ttsamd_copy_strided(segs, ns, stream));
        return TTSAMD_OK;
    });
}

extern "C" int ttsamd_vits_destroy(void *handle)
{
    return abi_guard("vits_destroy", [&]() -> int {
        if (!handle) return TTSAMD_OK;
        (void)hipDeviceSynchronize();
        delete as_model(handle);
        return TTSAMD_OK;
    });
}
