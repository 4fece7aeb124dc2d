// Hostile-input and memory-safety pass over the model-level C ABI (include/tts_amd.h: ttsamd_hifigan_*, ttsamd_vits_*, ttsamd_glowtts_*)
// on the CPU, under -fsanitize=address,undefined.  Built by tests/test_host_cpu.py from the REAL handle sources
// (csrc/hifigan_model.hip, vits_model.hip, glow_model.hip, model_common.hip, model_layers.hip, pack_host.cpp — compiled as plain C++)
// against tests/native/hip_stub (a malloc-backed "device") and tests/native/kernel_stubs.cpp (stand-ins for the kernel launches that
// read / write exactly the extents their arguments declare).  What it proves:
//   * "never throws / aborts across the ABI" (include/tts_amd.h:7): malformed configs (0 / 13 upsample layers, kernel < stride, odd
//     channels, bad flow counts), wrong-shape / duplicate / absurd loads, finalize without weights / twice, forward / encode / decode
//     before finalize or out of order, NULL pointers, a failing device allocation — all come back as negative codes with a message;
//   * the handles' workspace arithmetic, weight-image sizes and pointer bookkeeping are exact: every launch's declared extents land
//     inside exactly-sized heap buffers (ragged and plain batches, two request shapes in a row, shrinking and growing), no leaks.
// The numerics of the same code run on the GPU (tests/test_hifigan_gpu.py, tests/test_native_models_gpu.py).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/tts_amd.h"

extern long g_hip_stub_fail_after, g_hip_stub_live, g_stub_launches;

#define REQUIRE(c)                                                                                            \
    do {                                                                                                      \
        if (!(c)) {                                                                                           \
            fprintf(stderr, "FAILED %s:%d: %s   (last error: %s)\n", __FILE__, __LINE__, #c, ttsamd_last_error()); \
            return 1;                                                                                         \
        }                                                                                                     \
    } while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static float frand()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (float)((double)(rng_state >> 11) / 9007199254740992.0 - 0.5);
}

struct Sd {
    std::map<std::string, std::pair<std::vector<int64_t>, std::vector<float>>> t;
    void add(const std::string &name, std::vector<int64_t> shape, float scale = 1.f, float offset = 0.f)
    {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        std::vector<float> d((size_t)n);
        for (auto &v : d) v = frand() * scale + offset;
        t[name] = {shape, d};
    }
    void conv(const std::string &name, int co, int ci, int k, bool wn = false, bool bias = true)
    {
        if (wn) {
            add(name + ".parametrizations.weight.original0", {co, 1, 1}, 0.2f, 1.f);
            add(name + ".parametrizations.weight.original1", {co, ci, k});
        } else {
            add(name + ".weight", {co, ci, k});
        }
        if (bias) add(name + ".bias", {co});
    }
    void norm(const std::string &name, int c)
    {
        add(name + ".gamma", {c}, 0.1f, 1.f);
        add(name + ".beta", {c}, 0.1f);
    }
};

typedef int (*load_fn)(void *, const char *, const float *, const int64_t *, int);
static int load_all(load_fn f, void *h, const Sd &sd, const char *skip = nullptr)
{
    for (const auto &kv : sd.t) {
        if (skip && kv.first.find(skip) != std::string::npos) continue;
        const int rc = f(h, kv.first.c_str(), kv.second.second.data(), kv.second.first.data(), (int)kv.second.first.size());
        if (rc) return rc;
    }
    return 0;
}

static ttsamd_hifigan_config voc_cfg(int in_ch, int c0, int pad)
{
    ttsamd_hifigan_config c;
    memset(&c, 0, sizeof(c));
    c.in_channels = in_ch;
    c.out_channels = 1;
    c.resblock_type = 1;
    c.num_kernels = 3;
    const int ks[3] = {3, 7, 11}, ds[3] = {1, 3, 5};
    for (int j = 0; j < 3; ++j) {
        c.resblock_kernel_sizes[j] = ks[j];
        c.num_dilations[j] = 3;
        for (int d = 0; d < 3; ++d) c.resblock_dilation_sizes[j][d] = ds[d];
    }
    c.num_upsamples = 4;
    const int up[4] = {8, 8, 2, 2}, uk[4] = {16, 16, 4, 4};
    for (int i = 0; i < 4; ++i) c.upsample_factors[i] = up[i], c.upsample_kernel_sizes[i] = uk[i];
    c.upsample_initial_channel = c0;
    c.inference_padding = pad;
    c.precision = 0;
    return c;
}

static void voc_weights(Sd &sd, const std::string &p, const ttsamd_hifigan_config &c, bool wn)
{
    sd.conv(p + "conv_pre", c.upsample_initial_channel, c.in_channels, 7, wn);
    int ch = c.upsample_initial_channel;
    for (int i = 0; i < c.num_upsamples; ++i) {
        // ConvTranspose1d weight [c_in, c_out, k]
        if (wn) {
            sd.add(p + "ups." + std::to_string(i) + ".parametrizations.weight.original0", {ch, 1, 1}, 0.2f, 1.f);
            sd.add(p + "ups." + std::to_string(i) + ".parametrizations.weight.original1", {ch, ch / 2, c.upsample_kernel_sizes[i]});
        } else {
            sd.add(p + "ups." + std::to_string(i) + ".weight", {ch, ch / 2, c.upsample_kernel_sizes[i]});
        }
        sd.add(p + "ups." + std::to_string(i) + ".bias", {ch / 2});
        ch /= 2;
        for (int j = 0; j < c.num_kernels; ++j)
            for (int d = 0; d < c.num_dilations[j]; ++d) {
                const std::string rp = p + "resblocks." + std::to_string(i * c.num_kernels + j) + ".";
                sd.conv(rp + "convs1." + std::to_string(d), ch, ch, c.resblock_kernel_sizes[j], wn);
                sd.conv(rp + "convs2." + std::to_string(d), ch, ch, c.resblock_kernel_sizes[j], wn);
            }
    }
    sd.conv(p + "conv_post", 1, ch, 7, wn, false);
}

static int test_hifigan()
{
    void *h = nullptr;
    ttsamd_hifigan_config c = voc_cfg(80, 32, 5);
    // ---- malformed configurations --------------------------------------------------------------------------------------------
    REQUIRE(ttsamd_hifigan_create(nullptr, &h) < 0 && ttsamd_hifigan_create(&c, nullptr) < 0);
    for (int bad = 0; bad < 8; ++bad) {
        ttsamd_hifigan_config b = c;
        switch (bad) {
            case 0: b.num_upsamples = 0; break;
            case 1: b.num_upsamples = 13; break;
            case 2: b.upsample_kernel_sizes[1] = 4; break;           // kernel < stride
            case 3: b.upsample_initial_channel = 30; break;          // cannot be halved four times
            case 4: b.num_kernels = 9; break;
            case 5: b.resblock_kernel_sizes[0] = 4; break;           // even resblock kernel
            case 6: b.precision = 7; break;
            case 7: b.num_dilations[2] = 0; break;
        }
        void *hb = nullptr;
        REQUIRE(ttsamd_hifigan_create(&b, &hb) < 0 && hb == nullptr && strlen(ttsamd_last_error()) > 0);
    }
    REQUIRE(ttsamd_hifigan_create(&c, &h) == 0 && h);
    // ---- before finalize ------------------------------------------------------------------------------------------------------
    std::vector<float> mel((size_t)2 * 80 * 9, 0.1f), wav((size_t)2 * (9 + 10) * 256);
    REQUIRE(ttsamd_hifigan_forward(h, mel.data(), 2, 9, nullptr, wav.data(), 0, nullptr) < 0);
    REQUIRE(ttsamd_hifigan_finalize(h) < 0);                         // nothing loaded
    REQUIRE(ttsamd_hifigan_forward(h, mel.data(), 2, 9, nullptr, wav.data(), 0, nullptr) < 0);
    // ---- hostile loads --------------------------------------------------------------------------------------------------------
    const float one = 1.f;
    int64_t shp[4] = {1, 1, 1, 1};
    REQUIRE(ttsamd_hifigan_load(h, nullptr, &one, shp, 1) < 0 && ttsamd_hifigan_load(h, "x", nullptr, shp, 1) < 0 && ttsamd_hifigan_load(h, "x", &one, nullptr, 1) < 0);
    REQUIRE(ttsamd_hifigan_load(h, "x", &one, shp, 0) < 0 && ttsamd_hifigan_load(h, "x", &one, shp, 5) < 0);
    int64_t neg[2] = {-4, 3}, huge[3] = {(int64_t)1 << 30, (int64_t)1 << 30, 8};
    REQUIRE(ttsamd_hifigan_load(h, "x", &one, neg, 2) < 0 && ttsamd_hifigan_load(h, "x", &one, huge, 3) < 0);
    Sd sd;
    voc_weights(sd, "", c, true);
    REQUIRE(load_all(ttsamd_hifigan_load, h, sd, "resblocks.7.convs2.1") == 0);
    REQUIRE(ttsamd_hifigan_finalize(h) < 0 && strstr(ttsamd_last_error(), "resblocks.7.convs2.1"));        // a missing layer is named
    REQUIRE(ttsamd_hifigan_forward(h, mel.data(), 2, 9, nullptr, wav.data(), 0, nullptr) < 0);           // a failed finalize leaves no half-packed model behind
    {   // wrong-shape tensor
        Sd bad = sd;
        bad.t["conv_post.parametrizations.weight.original1"].first = {1, 8, 7};
        bad.t["conv_post.parametrizations.weight.original1"].second.resize(56);
        REQUIRE(load_all(ttsamd_hifigan_load, h, bad) == 0 && ttsamd_hifigan_finalize(h) < 0);
    }
    REQUIRE(load_all(ttsamd_hifigan_load, h, sd) == 0);
    REQUIRE(load_all(ttsamd_hifigan_load, h, sd) == 0);              // duplicate loads replace
    REQUIRE(ttsamd_hifigan_finalize(h) == 0);
    REQUIRE(ttsamd_hifigan_finalize(h) == 0);                        // twice: keeps the packed model
    // ---- forward: plain, ragged, shrinking and growing shapes; every launch's extents inside exactly-sized buffers --------------
    REQUIRE(ttsamd_hifigan_output_samples(h, 9) == (9 + 10) * 256);
    const long before = g_stub_launches;
    REQUIRE(ttsamd_hifigan_forward(h, mel.data(), 2, 9, nullptr, wav.data(), 0, nullptr) == 0);
    REQUIRE(g_stub_launches - before > 40);
    int64_t lens[2] = {9, 4};
    REQUIRE(ttsamd_hifigan_forward(h, mel.data(), 2, 9, lens, wav.data(), 0, nullptr) == 0);
    REQUIRE(ttsamd_hifigan_forward(h, mel.data(), 1, 3, nullptr, wav.data(), 0, nullptr) == 0);
    // MRF branch streams off (a host with several requests in flight): one set of ping-pong buffers, one stream
    REQUIRE(ttsamd_hifigan_set_option(h, TTSAMD_HIFIGAN_OPT_CONCURRENT_BRANCHES, 0) == 0 && ttsamd_hifigan_set_option(h, 99, 0) < 0);
    REQUIRE(ttsamd_hifigan_forward(h, mel.data(), 2, 9, lens, wav.data(), 0, nullptr) == 0);
    REQUIRE(ttsamd_hifigan_set_option(h, TTSAMD_HIFIGAN_OPT_CONCURRENT_BRANCHES, 1) == 0);
    std::vector<float> mel2((size_t)3 * 80 * 17, 0.2f), wav2((size_t)3 * (17 + 10) * 256);
    REQUIRE(ttsamd_hifigan_forward(h, mel2.data(), 3, 17, nullptr, wav2.data(), 0, nullptr) == 0);
    REQUIRE(ttsamd_hifigan_forward(h, nullptr, 1, 3, nullptr, wav.data(), 0, nullptr) < 0 && ttsamd_hifigan_forward(h, mel.data(), 1, 0, nullptr, wav.data(), 0, nullptr) < 0);
    REQUIRE(ttsamd_hifigan_forward_ex(h, mel.data(), 2, 9, lens, mel.data(), wav.data(), 0, nullptr) < 0);      // lengths and in_mask are alternatives
    REQUIRE(ttsamd_hifigan_forward(h, mel.data(), 2, 9, nullptr, wav.data(), 1, nullptr) < 0);           // graphs: unsupported by the stub runtime -> a code, no leak
    // ---- a device allocation that fails in the middle of a (re-)finalize / a workspace growth -----------------------------------
    REQUIRE(load_all(ttsamd_hifigan_load, h, sd) == 0);
    g_hip_stub_fail_after = 17;
    REQUIRE(ttsamd_hifigan_finalize(h) < 0);
    g_hip_stub_fail_after = -1;
    REQUIRE(ttsamd_hifigan_forward(h, mel.data(), 2, 9, nullptr, wav.data(), 0, nullptr) < 0);
    REQUIRE(load_all(ttsamd_hifigan_load, h, sd) == 0 && ttsamd_hifigan_finalize(h) == 0);
    std::vector<float> mel3((size_t)4 * 80 * 33, 0.3f), wav3((size_t)4 * (33 + 10) * 256);
    g_hip_stub_fail_after = 0;
    REQUIRE(ttsamd_hifigan_forward(h, mel3.data(), 4, 33, nullptr, wav3.data(), 0, nullptr) < 0);      // the workspace cannot grow
    REQUIRE(ttsamd_hifigan_forward(h, mel3.data(), 4, 33, nullptr, wav3.data(), 0, nullptr) == 0);     // ... and the handle is still usable
    REQUIRE(ttsamd_hifigan_destroy(h) == 0 && ttsamd_hifigan_destroy(nullptr) == 0);
    return 0;
}

static void transformer_weights(Sd &sd, const std::string &p, int H, int F, int layers, int k, int window, int heads)
{
    for (int i = 0; i < layers; ++i) {
        const std::string a = p + "attn_layers." + std::to_string(i) + ".", f = p + "ffn_layers." + std::to_string(i) + ".";
        for (const char *n : {"conv_q", "conv_k", "conv_v", "conv_o"}) sd.conv(a + n, H, H, 1);
        if (window > 0) {
            sd.add(a + "emb_rel_k", {1, 2 * window + 1, H / heads});
            sd.add(a + "emb_rel_v", {1, 2 * window + 1, H / heads});
        }
        sd.norm(p + "norm_layers_1." + std::to_string(i), H);
        sd.conv(f + "conv_1", F, H, k);
        sd.conv(f + "conv_2", H, F, k);
        sd.norm(p + "norm_layers_2." + std::to_string(i), H);
    }
}
static void dds_weights(Sd &sd, const std::string &p, int C)
{
    for (int i = 0; i < 3; ++i) {
        const std::string si = std::to_string(i);
        sd.add(p + "convs_sep." + si + ".weight", {C, 1, 3});
        sd.add(p + "convs_sep." + si + ".bias", {C});
        sd.conv(p + "convs_1x1." + si, C, C, 1);
        sd.norm(p + "norms_1." + si, C);
        sd.norm(p + "norms_2." + si, C);
    }
}
static void wn_weights(Sd &sd, const std::string &p, int H, int k, int layers)
{
    for (int i = 0; i < layers; ++i) {
        sd.conv(p + "in_layers." + std::to_string(i), 2 * H, H, k, true);
        sd.conv(p + "res_skip_layers." + std::to_string(i), i < layers - 1 ? 2 * H : H, H, 1, true);
    }
}

static int test_vits(bool use_sdp)
{
    ttsamd_vits_config c;
    memset(&c, 0, sizeof(c));
    const int H = 32;
    c.num_chars = 50;
    c.hidden_channels = H;
    c.hidden_channels_ffn_text_encoder = 48;
    c.num_heads_text_encoder = 2;
    c.num_layers_text_encoder = 3;          // odd: the result lands in the second ping-pong buffer
    c.kernel_size_text_encoder = 3;
    c.kernel_size_flow = 5;
    c.dilation_rate_flow = 1;
    c.num_layers_flow = 2;
    c.num_flows = 4;
    c.use_sdp = use_sdp;
    c.inference_noise_scale = 0.667f;
    c.inference_noise_scale_dp = use_sdp ? 0.8f : 1.f;
    c.length_scale = 1.f;
    c.decoder = voc_cfg(H, 32, 0);
    void *h = nullptr;
    for (int bad = 0; bad < 7; ++bad) {
        ttsamd_vits_config b = c;
        switch (bad) {
            case 0: b.num_flows = 3; break;
            case 1: b.hidden_channels = 33; break;
            case 2: b.num_heads_text_encoder = 5; break;
            case 3: b.decoder.inference_padding = 5; break;
            case 4: b.decoder.in_channels = 80; break;
            case 5: b.decoder.num_upsamples = 13; break;
            case 6: b.length_scale = 0.f; break;
        }
        void *hb = nullptr;
        REQUIRE(ttsamd_vits_create(&b, &hb) < 0 && hb == nullptr);
    }
    REQUIRE(ttsamd_vits_create(&c, &h) == 0 && h);
    const int B = 2, T = 7;
    std::vector<int64_t> x((size_t)B * T), xl = {7, 4}, ylh(B);
    for (size_t i = 0; i < x.size(); ++i) x[i] = (int64_t)(i * 7 % 50);
    std::vector<float> ndp((size_t)B * 2 * T, 0.3f);
    int32_t td = 0;
    REQUIRE(ttsamd_vits_encode(h, x.data(), xl.data(), B, T, ndp.data(), nullptr, 0, ylh.data(), &td, 0, nullptr) < 0);      // before finalize
    REQUIRE(ttsamd_vits_finalize(h) < 0);
    Sd sd;
    sd.add("text_encoder.emb.weight", {50, H});
    transformer_weights(sd, "text_encoder.encoder.", H, 48, 3, 3, 4, 2);
    sd.conv("text_encoder.proj", 2 * H, H, 1);
    if (use_sdp) {
        sd.conv("duration_predictor.pre", 192, H, 1);
        dds_weights(sd, "duration_predictor.convs.", 192);
        sd.conv("duration_predictor.proj", 192, 192, 1);
        sd.add("duration_predictor.flows.0.translation", {2, 1});
        sd.add("duration_predictor.flows.0.log_scale", {2, 1});
        for (int i = 1; i <= 4; ++i) {
            const std::string q = "duration_predictor.flows." + std::to_string(i) + ".";
            sd.conv(q + "pre", 192, 1, 1);
            dds_weights(sd, q + "convs.", 192);
            sd.conv(q + "proj", 29, 192, 1);
        }
    } else {
        sd.conv("duration_predictor.conv_1", 256, H, 3);
        sd.conv("duration_predictor.conv_2", 256, 256, 3);
        sd.norm("duration_predictor.norm_1", 256);
        sd.norm("duration_predictor.norm_2", 256);
        sd.conv("duration_predictor.proj", 1, 256, 1);
    }
    for (int i = 0; i < 4; ++i) {
        const std::string q = "flow.flows." + std::to_string(i) + ".";
        sd.conv(q + "pre", H, H / 2, 1);
        sd.conv(q + "post", H / 2, H, 1);
        wn_weights(sd, q + "enc.", H, 5, 2);
    }
    voc_weights(sd, "waveform_decoder.", c.decoder, false);
    sd.add("disc.whatever", {3});                               // ignored
    sd.add("posterior_encoder.pre.weight", {4, 4, 1});          // ignored
    REQUIRE(load_all(ttsamd_vits_load, h, sd, "flow.flows.2.enc.in_layers.1") == 0);
    REQUIRE(ttsamd_vits_finalize(h) < 0 && strstr(ttsamd_last_error(), "flow.flows.2.enc.in_layers.1"));
    REQUIRE(ttsamd_vits_encode(h, x.data(), xl.data(), B, T, ndp.data(), nullptr, 0, ylh.data(), &td, 0, nullptr) < 0);
    REQUIRE(load_all(ttsamd_vits_load, h, sd) == 0 && ttsamd_vits_finalize(h) == 0 && ttsamd_vits_finalize(h) == 0);
    REQUIRE(ttsamd_vits_hop_length(h) == 256);
    REQUIRE(ttsamd_vits_set_option(h, TTSAMD_HIFIGAN_OPT_CONCURRENT_BRANCHES, 0) == 0 && ttsamd_vits_set_option(h, TTSAMD_HIFIGAN_OPT_CONCURRENT_BRANCHES, 1) == 0);
    // decode before encode; NULL pointers; the SDP without its noise; a token outside the table (caught by the embed stand-in)
    ttsamd_vits_outputs o;
    memset(&o, 0, sizeof(o));
    std::vector<float> nz(1, 0.f);
    REQUIRE(ttsamd_vits_decode(h, nz.data(), &o, 0, nullptr) < 0);
    REQUIRE(ttsamd_vits_encode(h, nullptr, xl.data(), B, T, ndp.data(), nullptr, 0, ylh.data(), &td, 0, nullptr) < 0);
    REQUIRE(ttsamd_vits_encode(h, x.data(), xl.data(), 0, T, ndp.data(), nullptr, 0, ylh.data(), &td, 0, nullptr) < 0);
    if (use_sdp) REQUIRE(ttsamd_vits_encode(h, x.data(), xl.data(), B, T, nullptr, nullptr, 0, ylh.data(), &td, 0, nullptr) < 0);
    {
        std::vector<int64_t> xb = x;
        xb[3] = 50;
        REQUIRE(ttsamd_vits_encode(h, xb.data(), xl.data(), B, T, ndp.data(), nullptr, 0, ylh.data(), &td, 0, nullptr) < 0);
        REQUIRE(ttsamd_vits_decode(h, nz.data(), &o, 0, nullptr) < 0);     // a failed encode leaves no request behind
    }
    // a whole request, every output wanted / only the waveform wanted; then a larger request (workspaces grow), then a smaller one
    for (int round = 0; round < 3; ++round) {
        const int b = round == 1 ? 3 : 2, t = round == 1 ? 11 : (round == 2 ? 5 : 7);
        std::vector<int64_t> xx((size_t)b * t), ll((size_t)b), yl((size_t)b);
        for (size_t i = 0; i < xx.size(); ++i) xx[i] = (int64_t)(i % 50);
        for (int i = 0; i < b; ++i) ll[i] = t - 2 * i;
        std::vector<float> nd((size_t)b * 2 * t, 0.1f);
        REQUIRE(ttsamd_vits_encode(h, xx.data(), ll.data(), b, t, nd.data(), nullptr, 0, yl.data(), &td, 0, nullptr) == 0);
        REQUIRE(td > 0 && yl[0] == td);
        std::vector<float> noise((size_t)b * H * td, 0.2f), wavv((size_t)b * td * 256), attn((size_t)b * t * td), dur((size_t)b * t), z((size_t)b * H * td), zp(z.size()),
            mp(z.size()), lp(z.size()), ym((size_t)b * td), lw((size_t)b * t), xh((size_t)b * H * t);
        std::vector<int64_t> yl_dev((size_t)b);
        memset(&o, 0, sizeof(o));
        o.wav = wavv.data();
        if (round != 2) {
            o.alignments = attn.data(), o.durations = dur.data(), o.z = z.data(), o.z_p = zp.data(), o.m_p = mp.data(), o.logs_p = lp.data(), o.y_mask = ym.data();
            o.y_lengths = yl_dev.data(), o.logw = lw.data(), o.x_hidden = xh.data();
        }
        REQUIRE(ttsamd_vits_decode(h, noise.data(), &o, 0, nullptr) == 0);
        if (round != 2) REQUIRE(yl_dev[0] == td && dur[0] > 0.f);
        o.wav = nullptr;
        REQUIRE(ttsamd_vits_decode(h, noise.data(), &o, 0, nullptr) < 0);
        // injected durations, predictor skipped
        std::vector<float> din((size_t)b * t, 3.f);
        REQUIRE(ttsamd_vits_encode(h, xx.data(), ll.data(), b, t, nullptr, din.data(), 0, yl.data(), &td, 0, nullptr) == 0 && td == 3 * t);
    }
    // a device allocation failing while the workspace grows: a code, and the handle stays usable
    {
        const int b = 4, t = 13;
        std::vector<int64_t> xx((size_t)b * t, 1), ll((size_t)b, t), yl((size_t)b);
        std::vector<float> nd((size_t)b * 2 * t, 0.1f);
        g_hip_stub_fail_after = 0;
        REQUIRE(ttsamd_vits_encode(h, xx.data(), ll.data(), b, t, nd.data(), nullptr, 0, yl.data(), &td, 0, nullptr) < 0);
        REQUIRE(ttsamd_vits_encode(h, xx.data(), ll.data(), b, t, nd.data(), nullptr, 0, yl.data(), &td, 0, nullptr) == 0);
    }
    REQUIRE(ttsamd_vits_destroy(h) == 0 && ttsamd_vits_destroy(nullptr) == 0);
    return 0;
}

static int test_glow(bool mean_only, int window)
{
    ttsamd_glowtts_config c;
    memset(&c, 0, sizeof(c));
    const int H = 32, C = 8;
    c.num_chars = 40;
    c.hidden_channels_enc = H;
    c.hidden_channels_dec = 32;
    c.hidden_channels_dp = 24;
    c.out_channels = C;
    c.encoder_kernel_size = 3;
    c.encoder_num_layers = 2;
    c.encoder_num_heads = 2;
    c.encoder_hidden_channels_ffn = 40;
    c.encoder_rel_attn_window_size = window;
    c.encoder_layer_norm_type = window ? 2 : 1;
    c.use_encoder_prenet = 1;
    c.mean_only = mean_only;
    c.num_flow_blocks_dec = 3;
    c.kernel_size_dec = 5;
    c.dilation_rate = 1;
    c.num_block_layers = 2;
    c.num_splits = 4;
    c.num_squeeze = 2;
    c.inference_noise_scale = 0.3f;
    c.length_scale = 1.f;
    void *h = nullptr;
    for (int bad = 0; bad < 5; ++bad) {
        ttsamd_glowtts_config b = c;
        switch (bad) {
            case 0: b.num_splits = 2; break;
            case 1: b.out_channels = 7; break;
            case 2: b.encoder_num_heads = 3; break;
            case 3: b.encoder_layer_norm_type = 0; break;
            case 4: b.hidden_channels_dec = 30; break;
        }
        void *hb = nullptr;
        REQUIRE(ttsamd_glowtts_create(&b, &hb) < 0 && hb == nullptr);
    }
    REQUIRE(ttsamd_glowtts_create(&c, &h) == 0 && h);
    Sd sd;
    sd.add("encoder.emb.weight", {40, H});
    for (int i = 0; i < 3; ++i) {
        sd.conv("encoder.prenet.conv_layers." + std::to_string(i), H, H, 5);
        sd.norm("encoder.prenet.norm_layers." + std::to_string(i), H);
    }
    sd.conv("encoder.prenet.proj", H, H, 1);
    transformer_weights(sd, "encoder.encoder.", H, 40, 2, 3, window, 2);
    sd.conv("encoder.proj_m", C, H, 1);
    if (!mean_only) sd.conv("encoder.proj_s", C, H, 1);
    sd.conv("encoder.duration_predictor.conv_1", 24, H, 3);
    sd.conv("encoder.duration_predictor.conv_2", 24, 24, 3);
    sd.norm("encoder.duration_predictor.norm_1", 24);
    sd.norm("encoder.duration_predictor.norm_2", 24);
    sd.conv("encoder.duration_predictor.proj", 1, 24, 1);
    const int cq = C * 2;
    for (int b = 0; b < 3; ++b) {
        const std::string pa = "decoder.flows." + std::to_string(3 * b) + ".", pi = "decoder.flows." + std::to_string(3 * b + 1) + ".",
                          pc = "decoder.flows." + std::to_string(3 * b + 2) + ".";
        sd.add(pa + "bias", {1, cq, 1});
        sd.add(pa + "logs", {1, cq, 1});
        sd.add(pi + "weight", {4, 4}, 0.2f);
        for (int i = 0; i < 4; ++i) sd.t[pi + "weight"].second[(size_t)i * 5] += 1.f;
        sd.conv(pc + "start", 32, cq / 2, 1, true);
        wn_weights(sd, pc + "wn.", 32, 5, 2);
        sd.conv(pc + "end", cq, 32, 1);
    }
    const int B = 3, T = 9;
    std::vector<int64_t> x((size_t)B * T), xl = {9, 6, 2}, ylh(B);
    for (size_t i = 0; i < x.size(); ++i) x[i] = (int64_t)(i % 40);
    int32_t td = 0;
    REQUIRE(ttsamd_glowtts_encode(h, x.data(), xl.data(), B, T, nullptr, 0, ylh.data(), &td, 0, nullptr) < 0);
    REQUIRE(load_all(ttsamd_glowtts_load, h, sd, "prenet.norm_layers.1.beta") == 0 && ttsamd_glowtts_finalize(h) < 0 && strstr(ttsamd_last_error(), "norm_layers.1.beta"));
    {   // a singular InvConvNear weight is a code, not a division by zero
        Sd bad = sd;
        std::fill(bad.t["decoder.flows.4.weight"].second.begin(), bad.t["decoder.flows.4.weight"].second.end(), 0.f);
        REQUIRE(load_all(ttsamd_glowtts_load, h, bad) == 0 && ttsamd_glowtts_finalize(h) < 0 && strstr(ttsamd_last_error(), "singular"));
    }
    REQUIRE(load_all(ttsamd_glowtts_load, h, sd) == 0 && ttsamd_glowtts_finalize(h) == 0 && ttsamd_glowtts_finalize(h) == 0);
    ttsamd_glowtts_outputs o;
    memset(&o, 0, sizeof(o));
    REQUIRE(ttsamd_glowtts_decode(h, nullptr, &o, nullptr) < 0);
    for (int round = 0; round < 3; ++round) {
        const int ragged = round == 1;
        REQUIRE(ttsamd_glowtts_encode(h, x.data(), xl.data(), B, T, nullptr, ragged, ylh.data(), &td, 0, nullptr) == 0 && td > 0);
        const int ty = td / 2 * 2;
        std::vector<float> noise((size_t)B * C * td, 0.1f), mel((size_t)B * C * ty), ym((size_t)B * C * td), yls(ym.size()), attn((size_t)B * T * td), dl((size_t)B * T), tdl(dl.size()),
            dur(dl.size());
        std::vector<int64_t> yl((size_t)B);
        memset(&o, 0, sizeof(o));
        o.mel = mel.data();
        if (round != 2) o.y_mean = ym.data(), o.y_log_scale = yls.data(), o.alignments = attn.data(), o.durations_log = dl.data(), o.total_durations_log = tdl.data(), o.durations = dur.data(), o.y_lengths = yl.data();
        REQUIRE(ttsamd_glowtts_decode(h, nullptr, &o, nullptr) < 0);                 // inference_noise_scale != 0 needs the draw
        REQUIRE(ttsamd_glowtts_decode(h, noise.data(), &o, nullptr) == 0);
    }
    REQUIRE(ttsamd_glowtts_destroy(h) == 0 && ttsamd_glowtts_destroy(nullptr) == 0);
    return 0;
}

int main()
{
    if (test_hifigan()) return 1;
    if (test_vits(true) || test_vits(false)) return 1;
    if (test_glow(true, 0) || test_glow(false, 4)) return 1;
    if (g_hip_stub_live != 0) {
        fprintf(stderr, "FAILED: %ld device / pinned allocations were never freed\n", g_hip_stub_live);
        return 1;
    }
    printf("handles_driver: ok (%ld stub launches)\n", g_stub_launches);
    return 0;
}
