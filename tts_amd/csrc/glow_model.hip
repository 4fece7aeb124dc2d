// Model-level C ABI of Glow-TTS ("mi355_glowtts_{...}"): GlowTTS.inference of the single-speaker model
// (TTS/tts/models/glow_tts.py:341-374) behind ONE handle, in two calls around the request's one host wait (include/tts_amd.h):
//
//   encode  embed -> prenet 3 x [conv k5 -> LayerNorm -> ReLU] + 1x1 residual -> rel-pos transformer -> proj_m (-> proj_s)
//           -> DurationPredictor -> w = (exp(logw) - 1) mask length_scale, max(ceil, 1), cumsum, y_lengths      encoder.py:143-179, glow_tts.py:350-352
//   decode  y_mean / y_log_scale gathered along the path (+ noise), generate_path                               glow_tts.py:137-148,361
//           -> squeeze -> 12 x reversed [CouplingBlock^-1 (start 1x1, WN, end 1x1 + affine coupling), InvConvNear^-1, ActNorm^-1] -> unsqueeze
//                                                                                                                decoder.py:113-141, glow.py:70-233
// Every launch goes through the kernel-level ABI with the arguments the Python host (tts_amd/glow_tts.py, layers.py) passes.
#include "model_layers.h"

using namespace ttsamd;
using namespace ttsamd::model;

namespace {

constexpr const char *kWho = "glowtts";

struct Block {
    PackedConv start, end;
    Wn wn;
    DevBuf w_inv, an_bias, an_logs, mix;
};

struct Model {
    ttsamd_glowtts_config cfg{};
    TensorMap tensors;
    bool finalized = false;
    DevBuf emb;
    bool has_prenet = false;
    PackedConv pre_conv[3], pre_proj;
    Norm pre_norm[3];
    Transformer enc;
    PackedConv proj_m, proj_s;
    Dp dp;
    std::vector<std::unique_ptr<Block>> blocks;
    bool fuse_mix = false;
    DevBuf work, work2;
    int64_t *host_len = nullptr;
    int host_len_cap = 0;
    struct Req {
        bool valid = false;
        int B = 0, T = 0, t_dec = 0;
        const float *x_mask = nullptr, *o_mean = nullptr, *o_logs = nullptr, *logw = nullptr;
        float *w_ceil = nullptr;
        int32_t *cum = nullptr;
        int64_t *y_lengths = nullptr;
    } req;
    GraphCache front_graphs;
    ~Model()
    {
        if (host_len) (void)hipHostFree(host_len);
    }
};

Model *as_model(void *h) { return static_cast<Model *>(h); }

// 4x4 inverse in double precision (store_inverse, glow.py:139-141, when the checkpoint carries the weight only)
bool invert4(const float *w, float *out)
{
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = w[i * 4 + j];
            a[i][4 + j] = i == j ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r)
            if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        if (a[piv][col] == 0.0) return false;
        if (piv != col)
            for (int j = 0; j < 8; ++j) std::swap(a[piv][j], a[col][j]);
        const double d = a[col][col];
        for (int j = 0; j < 8; ++j) a[col][j] /= d;
        for (int r = 0; r < 4; ++r)
            if (r != col) {
                const double f = a[r][col];
                for (int j = 0; j < 8; ++j) a[r][j] -= f * a[col][j];
            }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out[i * 4 + j] = (float)a[i][4 + j];
    return true;
}

int finalize(Model &m)
{
    const ttsamd_glowtts_config &c = m.cfg;
    const int H = c.hidden_channels_enc, C = c.out_channels;
    m.blocks.clear();
    m.front_graphs.clear();
    m.req.valid = false;
    RC(upload_named(m.tensors, kWho, "encoder.emb.weight", (int64_t)c.num_chars * H, m.emb));
    m.has_prenet = c.use_encoder_prenet != 0;
    if (m.has_prenet) {     // ResidualConv1dLayerNormBlock(hidden, hidden, hidden, kernel 5, 3 layers), glow.py:11-67
        for (int i = 0; i < 3; ++i) {
            const std::string si = std::to_string(i);
            RC(pack_named_conv(m.tensors, kWho, "encoder.prenet.conv_layers." + si, m.pre_conv[i], H, H, 5, 1));
            RC(upload_norm(m.tensors, kWho, "encoder.prenet.norm_layers." + si, H, 1e-4f, m.pre_norm[i]));
        }
        RC(pack_named_conv(m.tensors, kWho, "encoder.prenet.proj", m.pre_proj, H, H, 1, 1));
    }
    RC(build_transformer(m.tensors, kWho, "encoder.encoder.", H, c.encoder_hidden_channels_ffn, c.encoder_num_heads, c.encoder_num_layers,
                         c.encoder_kernel_size, c.encoder_rel_attn_window_size, c.encoder_layer_norm_type == 2 ? 1e-5f : 1e-4f, m.enc));
    RC(pack_named_conv(m.tensors, kWho, "encoder.proj_m", m.proj_m, C, H, 1, 1));
    if (!c.mean_only) RC(pack_named_conv(m.tensors, kWho, "encoder.proj_s", m.proj_s, C, H, 1, 1));
    RC(build_dp(m.tensors, kWho, "encoder.duration_predictor.", H, c.hidden_channels_dp, m.dp));
    // decoder: flows.{3b} ActNorm, {3b+1} InvConvNear, {3b+2} CouplingBlock (decoder.py:84-111)
    const int cq = C * c.num_squeeze, half = cq / 2, ns = c.num_splits, hd = c.hidden_channels_dec;
    for (int b = 0; b < c.num_flow_blocks_dec; ++b) {
        auto blk = std::make_unique<Block>();
        const std::string pa = "decoder.flows." + std::to_string(3 * b) + ".", pi = "decoder.flows." + std::to_string(3 * b + 1) + ".",
                          pc = "decoder.flows." + std::to_string(3 * b + 2) + ".";
        RC(pack_named_conv(m.tensors, kWho, pc + "start", blk->start, hd, half, 1, 1));
        RC(build_wn(m.tensors, kWho, pc + "wn.", hd, c.kernel_size_dec, c.dilation_rate, c.num_block_layers, blk->wn));
        // end [cq, hidden, 1]: t rows | s rows -> paired-row packing for the COUPLE_AFFINE epilogue
        const HostTensor *we = nullptr, *be = nullptr;
        RC(need_tensor(m.tensors, kWho, pc + "end.weight", (int64_t)cq * hd, &we));
        RC(need_tensor(m.tensors, kWho, pc + "end.bias", cq, &be));
        std::vector<float> wp, bp;
        pair_permute(we->data, be->data.data(), hd, half, half, wp, bp);
        RC(pack_conv(blk->end, kWho, wp.data(), bp.data(), (int)bp.size(), hd, 1, 1, -1));
        std::vector<float> winv(ns * ns);
        if (const HostTensor *wi = find_tensor(m.tensors, pi + "weight_inv")) {
            if (wi->numel() != ns * ns) {
                set_error("glowtts: '%sweight_inv' is not %d x %d", pi.c_str(), ns, ns);
                return TTSAMD_ERR_INVALID;
            }
            winv = wi->data;
        } else {
            const HostTensor *w = nullptr;
            RC(need_tensor(m.tensors, kWho, pi + "weight", ns * ns, &w));
            if (!invert4(w->data.data(), winv.data())) {
                set_error("glowtts: '%sweight' is singular", pi.c_str());
                return TTSAMD_ERR_INVALID;
            }
        }
        const HostTensor *ab = nullptr, *al = nullptr;
        RC(need_tensor(m.tensors, kWho, pa + "bias", cq, &ab));
        RC(need_tensor(m.tensors, kWho, pa + "logs", cq, &al));
        RC(blk->w_inv.upload(winv.data(), winv.size() * 4));
        RC(blk->an_bias.upload(ab->data.data(), ab->data.size() * 4));
        RC(blk->an_logs.upload(al->data.data(), al->data.size() * 4));
        // the reverse block's InvConvNear^-1 + ActNorm^-1 parameters as ONE device block: they ride in the `end` conv's epilogue
        std::vector<float> mix(winv);
        mix.insert(mix.end(), ab->data.begin(), ab->data.end());
        mix.insert(mix.end(), al->data.begin(), al->data.end());
        RC(blk->mix.upload(mix.data(), mix.size() * 4));
        m.blocks.push_back(std::move(blk));
    }
    m.fuse_mix = ns == 4 && cq % 4 == 0 && half % 2 == 0;
    m.tensors.clear();
    m.finalized = true;
    return TTSAMD_OK;
}

// encoder + duration predictor (tts_amd/glow_tts.py: _front_eager); size pass / layout / launch as in vits_model.hip
int front(Model &m, Bump &ws, const int64_t *x, const int64_t *x_lengths, hipStream_t st, bool launch)
{
    const ttsamd_glowtts_config &c = m.cfg;
    const int B = m.req.B, T = m.req.T, H = c.hidden_channels_enc, F = c.encoder_hidden_channels_ffn, C = c.out_channels, D = c.hidden_channels_dp;
    Ctx cx{c.precision, reinterpret_cast<void *>(st), B, T};
    const size_t n = (size_t)B * T;
    float *x_mask = ws.take(n), *x0 = ws.take(n * H), *pc = ws.take(n * H), *pa = ws.take(n * H), *pb = ws.take(n * H);
    TransformerBufs tb;
    tb.take(ws, n, H, F);
    float *o_mean = ws.take(n * C), *o_logs = c.mean_only ? nullptr : ws.take(n * C), *logw = ws.take(n);
    float *d0 = ws.take(n * D), *d1 = ws.take(n * D), *d2 = ws.take(n * D), *d3 = ws.take(n * D);
    float *w_ceil = ws.take(n);
    int32_t *cum = ws.take_as<int32_t>(n);
    int64_t *ylen = ws.take_as<int64_t>(B);
    if (ws.dry) return TTSAMD_OK;
    m.req.x_mask = x_mask;
    m.req.o_mean = o_mean;
    m.req.o_logs = o_logs;
    m.req.logw = logw;
    m.req.w_ceil = w_ceil;
    m.req.cum = cum;
    m.req.y_lengths = ylen;
    if (!launch) return TTSAMD_OK;
    void *s = cx.s;
    ttsamd_conv1d_args a;
    RC(ttsamd_sequence_mask(x_mask, x_lengths, B, T, s));
    float *xin = m.has_prenet ? x0 : tb.xa;
    RC(ttsamd_embed(xin, x, m.emb.f(), x_mask, (float)std::sqrt((double)H), B, H, T, c.num_chars, s));     // emb(x) * sqrt(H) (encoder.py:156-160)
    if (m.has_prenet) {
        const float *h = x0;
        float *nb[2] = {pa, pb};
        for (int i = 0; i < 3; ++i) {
            fill_conv_args(cx.precision, a, m.pre_conv[i], h, H, T, pc, H, T, B);
            a.in_mask = x_mask;
            a.out_mask = x_mask;
            RC(conv(cx, a));
            RC(norm(cx, pc, nb[i & 1], H, T, m.pre_norm[i], TTSAMD_ACT_RELU));
            h = nb[i & 1];
        }
        fill_conv_args(cx.precision, a, m.pre_proj, h, H, T, tb.xa, H, T, B);           // x + proj(h), masked
        a.res = x0;
        a.res_bstride = (int64_t)H * T;
        a.res_rstride = T;
        a.out_mask = x_mask;
        RC(conv(cx, a));
    }
    float *xe = nullptr;
    RC(run_transformer(cx, m.enc, tb, x_mask, &xe));
    fill_conv_args(cx.precision, a, m.proj_m, xe, H, T, o_mean, C, T, B);
    a.out_mask = x_mask;
    RC(conv(cx, a));
    if (!c.mean_only) {
        fill_conv_args(cx.precision, a, m.proj_s, xe, H, T, o_logs, C, T, B);
        a.out_mask = x_mask;
        RC(conv(cx, a));
    }
    float *const db[4] = {d0, d1, d2, d3};
    return run_dp(cx, m.dp, xe, H, x_mask, db, logw);
}

int grow(DevBuf &buf, size_t bytes)
{
    if (bytes <= buf.bytes && buf.p) return TTSAMD_OK;
    TTSAMD_HIP(hipDeviceSynchronize());
    return buf.alloc(bytes);
}

}  // namespace

extern "C" int ttsamd_glowtts_create(const ttsamd_glowtts_config *cfg, void **handle_out)
{
    return abi_guard("glowtts_create", [&]() -> int {
        TTSAMD_CHECK_ARG(cfg && handle_out, "glowtts_create: NULL argument");
        const ttsamd_glowtts_config &c = *cfg;
        TTSAMD_CHECK_ARG(c.num_chars > 0 && c.hidden_channels_enc > 0 && c.hidden_channels_dec > 0 && c.hidden_channels_dp > 0 && c.out_channels > 0,
                         "glowtts_create: bad channel counts");
        TTSAMD_CHECK_ARG(c.encoder_num_heads > 0 && c.hidden_channels_enc % c.encoder_num_heads == 0 && c.hidden_channels_enc / c.encoder_num_heads <= 128,
                         "glowtts_create: hidden_channels_enc %d over %d heads (head size <= 128)", c.hidden_channels_enc, c.encoder_num_heads);
        TTSAMD_CHECK_ARG(c.encoder_num_layers >= 1 && c.encoder_num_layers <= 64 && c.encoder_kernel_size >= 1 && c.encoder_kernel_size <= 31 &&
                             c.encoder_hidden_channels_ffn > 0 && c.encoder_rel_attn_window_size >= 0 && c.encoder_rel_attn_window_size <= 64,
                         "glowtts_create: bad encoder_params");
        TTSAMD_CHECK_ARG(c.encoder_layer_norm_type == 1 || c.encoder_layer_norm_type == 2, "glowtts_create: layer_norm_type is 1 or 2");
        TTSAMD_CHECK_ARG(c.num_flow_blocks_dec >= 1 && c.num_flow_blocks_dec <= 64 && c.num_block_layers >= 1 && c.num_block_layers <= 32 &&
                             c.kernel_size_dec >= 1 && c.kernel_size_dec % 2 == 1 && c.dilation_rate >= 1,
                         "glowtts_create: bad decoder configuration");
        TTSAMD_CHECK_ARG(c.num_splits == 4, "glowtts_create: num_splits must be 4 (InvConvNear kernel)");
        TTSAMD_CHECK_ARG(c.num_squeeze >= 1 && c.num_squeeze <= 8 && (c.out_channels * c.num_squeeze) % (2 * c.num_splits) == 0,
                         "glowtts_create: out_channels * num_squeeze must split into coupling halves of whole %d-channel groups", c.num_splits);
        TTSAMD_CHECK_ARG(c.hidden_channels_dec % kPairRows == 0, "glowtts_create: the gate conv's paired rows need hidden_channels_dec %% %d == 0", kPairRows);
        TTSAMD_CHECK_ARG(c.length_scale > 0.f && c.precision >= 0 && c.precision <= 2, "glowtts_create: bad length_scale / precision");
        Model *m = new Model();
        m->cfg = c;
        *handle_out = m;
        return TTSAMD_OK;
    });
}

extern "C" int ttsamd_glowtts_load(void *handle, const char *name, const float *data, const int64_t *shape, int ndim)
{
    return abi_guard("glowtts_load", [&]() -> int {
        TTSAMD_CHECK_ARG(handle, "glowtts_load: NULL handle");
        Model &m = *as_model(handle);
        RC(load_tensor(m.tensors, kWho, name, data, shape, ndim));
        m.finalized = false;
        return TTSAMD_OK;
    });
}

extern "C" int ttsamd_glowtts_finalize(void *handle)
{
    return abi_guard("glowtts_finalize", [&]() -> int {
        TTSAMD_CHECK_ARG(handle, "glowtts_finalize: NULL handle");
        Model &m = *as_model(handle);
        if (m.finalized && m.tensors.empty()) return TTSAMD_OK;
        m.finalized = false;
        TTSAMD_HIP(hipDeviceSynchronize());
        return finalize(m);
    });
}

extern "C" int ttsamd_glowtts_encode(void *handle, const int64_t *x, const int64_t *x_lengths, int batch, int t_text, const float *durations_in,
                                     int ragged_exact, int64_t *y_lengths_host, int32_t *t_dec_out, int use_graph, void *stream)
{
    return abi_guard("glowtts_encode", [&]() -> int {
        TTSAMD_CHECK_ARG(handle && x && x_lengths && t_dec_out, "glowtts_encode: NULL argument");
        Model &m = *as_model(handle);
        TTSAMD_CHECK_ARG(m.finalized, "glowtts_encode: weights not loaded (ttsamd_glowtts_load ... ttsamd_glowtts_finalize)");
        TTSAMD_CHECK_ARG(batch >= 1 && batch <= 65535 && t_text >= 1, "glowtts_encode: bad shape [%d, %d]", batch, t_text);
        hipStream_t st = as_stream(stream);
        m.req = Model::Req();
        m.req.B = batch;
        m.req.T = t_text;
        Bump dry;
        RC(front(m, dry, x, x_lengths, st, false));
        if (dry.used > m.work.bytes) {
            m.front_graphs.clear();
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
            return front(m, ws, x, x_lengths, s2, launch);
        };
        const std::vector<const void *> kp = {x, x_lengths};
        const std::vector<int64_t> ki = {batch, t_text};
        GraphEntry *g = use_graph ? m.front_graphs.find(kp, ki, st) : nullptr;
        if (g) {
            RC(run(st, false));
            TTSAMD_HIP(hipGraphLaunch(g->exec, st));
        } else {
            RC(run(st, true));
            if (use_graph) RC(m.front_graphs.capture(kp, ki, st, [&](hipStream_t s2) { return run(s2, true); }));
        }
        // w = (exp(logw) - 1) * mask * length_scale, w_ceil = max(ceil(w), 1) (glow_tts.py:350-352): glow = 1; 2 = ragged-exact
        for (int i = 0; i < batch; ++i) m.host_len[i] = -1;
        RC(ttsamd_durations_ex(m.req.w_ceil, m.req.cum, m.req.y_lengths, m.host_len, durations_in ? nullptr : m.req.logw, durations_in, m.req.x_mask,
                               durations_in ? 1.0f : m.cfg.length_scale, durations_in ? 0 : (ragged_exact ? 2 : 1), t_text, batch, t_text, stream));
        volatile int64_t *hl = m.host_len;
        int64_t tmax = 0;
        for (int i = 0; i < batch; ++i) {
            unsigned long long spins = 0;
            while (hl[i] < 0) {
                if ((++spins & 0xFFFFF) == 0 && hipStreamQuery(st) == hipSuccess && hl[i] < 0) {
                    set_error("glowtts_encode: the durations kernel finished without publishing y_lengths");
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

extern "C" int ttsamd_glowtts_decode(void *handle, const float *noise, const ttsamd_glowtts_outputs *outp, void *stream)
{
    return abi_guard("glowtts_decode", [&]() -> int {
        TTSAMD_CHECK_ARG(handle && outp && outp->mel, "glowtts_decode: NULL argument (out, out->mel)");
        Model &m = *as_model(handle);
        TTSAMD_CHECK_ARG(m.finalized && m.req.valid, "glowtts_decode: no request in flight (ttsamd_glowtts_encode first)");
        const ttsamd_glowtts_config &c = m.cfg;
        const ttsamd_glowtts_outputs &o = *outp;
        TTSAMD_CHECK_ARG(noise || c.inference_noise_scale == 0.f, "glowtts_decode: inference_noise_scale %g needs a noise draw [batch, out_channels, t_dec]",
                         (double)c.inference_noise_scale);
        const int B = m.req.B, T = m.req.T, td = m.req.t_dec, C = c.out_channels, nsq = c.num_squeeze, tq = td / nsq, cq = C * nsq, half = cq / 2, hd = c.hidden_channels_dec;
        Ctx cx{c.precision, stream, B, T};
        const size_t nt = (size_t)B * td, nq = (size_t)B * tq;
        float *z_p, *m_p, *logs_p, *y_mask, *xq, *mq, *h, *acts, *skip;
        auto layout = [&](Bump &b) {
            z_p = b.take(nt * C);
            m_p = o.y_mean ? o.y_mean : b.take(nt * C);
            logs_p = o.y_log_scale ? o.y_log_scale : b.take(nt * C);
            y_mask = b.take(nt);
            xq = b.take(nq * cq + 4);
            mq = b.take(nq + 4);
            h = b.take(nq * hd + 4);
            acts = b.take(nq * hd + 4);
            skip = b.take(nq * hd + 4);
        };
        Bump ws;
        layout(ws);
        RC(grow(m.work2, ws.used));
        ws = Bump();
        ws.base = static_cast<unsigned char *>(m.work2.p);
        ws.dry = false;
        layout(ws);
        // y_mean / y_log_scale gathered along the path, z = (y_mean + exp(y_log_scale) * noise * noise_scale) * y_mask (glow_tts.py:137-148,361)
        RC(ttsamd_expand_prior_ex(z_p, nullptr, m_p, logs_p, y_mask, m.req.o_mean, m.req.o_logs, (int64_t)C * T, noise, m.req.cum, m.req.x_mask, m.req.y_lengths,
                                  c.inference_noise_scale, 1, 0, B, C, T, td, stream));
        if (o.alignments) RC(ttsamd_generate_path(o.alignments, m.req.cum, m.req.x_mask, m.req.y_lengths, B, T, td, stream));
        // Decoder.forward(reverse=True), decoder.py:113-141: in place on the squeezed buffer
        RC(ttsamd_glow_squeeze(xq, mq, z_p, y_mask, B, C, td, nsq, stream));
        ttsamd_conv1d_args a;
        for (int bi = (int)m.blocks.size() - 1; bi >= 0 && tq > 0; --bi) {
            const Block &blk = *m.blocks[bi];
            fill_conv_args(cx.precision, a, blk.start, xq, cq, tq, h, hd, tq, B);      // start(x0) * mask (x0 = first half)
            a.out_mask = mq;
            RC(conv(cx, a));
            RC(run_wn(cx, blk.wn, h, acts, skip, mq, hd, tq));
            fill_conv_args(cx.precision, a, blk.end, skip, hd, tq, xq + (size_t)half * tq, cq, tq, B);
            a.mode = m.fuse_mix ? TTSAMD_CONV_COUPLE_AFFINE_MIX : TTSAMD_CONV_COUPLE_AFFINE;
            fix_conv_mode(cx.precision, a, blk.end);
            a.res = xq + (size_t)half * tq;
            a.res_bstride = (int64_t)cq * tq;
            a.res_rstride = tq;
            a.out_mask = mq;
            a.split_row = half;
            if (m.fuse_mix) a.y2 = static_cast<float *>(blk.mix.p);
            RC(conv(cx, a));
            if (!m.fuse_mix) RC(ttsamd_glow_invconv_actnorm(xq, blk.w_inv.f(), blk.an_bias.f(), blk.an_logs.f(), mq, B, cq, tq, c.num_splits, 0, stream));
        }
        RC(ttsamd_glow_unsqueeze(o.mel, xq, mq, B, cq, tq, nsq, tq * nsq, stream));
        if (o.total_durations_log) RC(ttsamd_attn_durations(o.total_durations_log, m.req.cum, m.req.x_mask, m.req.y_lengths, B, T, stream));
        ttsamd_copy_seg segs[3];
        int ns = 0;
        auto add = [&](void *dst, const void *src, int64_t count, int bytes) {
            if (!dst || !src) return;
            ttsamd_copy_seg &g = segs[ns++];
            memset(&g, 0, sizeof(g));
            g.src = src;
            g.dst = dst;
            g.d0 = 1;
            g.d1 = 1;
            g.d2 = (int32_t)count;
            g.s2 = 1;
            g.t2 = 1;
            g.elem_bytes = bytes;
        };
        add(o.durations, m.req.w_ceil, (int64_t)B * T, 4);
        add(o.y_lengths, m.req.y_lengths, B, 8);
        add(o.durations_log, m.req.logw, (int64_t)B * T, 4);
        if (ns) RC(ttsamd_copy_strided(segs, ns, stream));
        return TTSAMD_OK;
    });
}

extern "C" int ttsamd_glowtts_destroy(void *handle)
{
    return abi_guard("glowtts_destroy", [&]() -> int {
        if (!handle) return TTSAMD_OK;
        (void)hipDeviceSynchronize();
        delete as_model(handle);
        return TTSAMD_OK;
    });
}
