"""Seeded synthetic checkpoints in the reference's parameter layout.

There is no network in the build/bench environment, hence no released checkpoints: benchmarks, smoke tests and
parity tests all run on random-init weights of the exact reference architectures.  The key names / shapes are
those of the reference modules' `state_dict()` (verified by tests/test_oracle_pin.py against the real modules
when /root/reference is present).  Layers the reference zero-initialises (networks.py:135-136,
stochastic_duration_predictor.py:117-118, glow.py:52-53,195-196) get small random values, otherwise every
coupling layer is the identity and parity would be vacuous.  Pure torch-CPU tensor construction; no kernels.
"""
import math

import torch

VITS_DEFAULTS = dict(  # VitsArgs, vits.py:544-600
    num_chars=100, hidden_channels=192, hidden_channels_ffn_text_encoder=768, num_heads_text_encoder=2,
    num_layers_text_encoder=6, kernel_size_text_encoder=3, kernel_size_flow=5, dilation_rate_flow=1,
    num_layers_flow=4, resblock_type_decoder="1", resblock_kernel_sizes_decoder=[3, 7, 11],
    resblock_dilation_sizes_decoder=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], upsample_rates_decoder=[8, 8, 2, 2],
    upsample_initial_channel_decoder=512, upsample_kernel_sizes_decoder=[16, 16, 4, 4], use_sdp=True,
    inference_noise_scale=0.667, length_scale=1.0, inference_noise_scale_dp=1.0, max_inference_len=None,
)



GLOW_DEFAULTS = dict(  # GlowTTSConfig, glow_tts_config.py:101-152
    num_chars=130, hidden_channels_enc=192, hidden_channels_dec=192, hidden_channels_dp=256, out_channels=80,
    num_flow_blocks_dec=12, kernel_size_dec=5, dilation_rate=1, num_block_layers=4, num_splits=4, num_squeeze=2,
    sigmoid_scale=False, mean_only=True, use_encoder_prenet=True, inference_noise_scale=0.0, length_scale=1.0,
    encoder_params=dict(kernel_size=3, num_layers=6, num_heads=2, hidden_channels_ffn=768,
                        rel_attn_window_size=None, layer_norm_type="1"),
)



def _conv(sd, name, cout, cin, k, gen, wn=False, bias=True, std=None, transposed=False):
    fan_in = cin * k
    std = std if std is not None else 1.0 / math.sqrt(fan_in)
    shape = (cin, cout, k) if transposed else (cout, cin, k)
    v = torch.randn(shape, generator=gen) * std
    if wn:
        n0 = shape[0]
        g = v.reshape(n0, -1).norm(dim=1).reshape(n0, 1, 1) * (0.8 + 0.4 * torch.rand(n0, 1, 1, generator=gen))
        sd[name + ".parametrizations.weight.original0"] = g
        sd[name + ".parametrizations.weight.original1"] = v
    else:
        sd[name + ".weight"] = v
    if bias:
        sd[name + ".bias"] = torch.randn(cout, generator=gen) * 0.02


def make_hifigan_state(cfg, in_channels, seed=1234, prefix="", weight_norm=True, pre_wn=True, post_wn=True,
                       post_bias=True, out_channels=1):
    """Random HifiganGenerator state_dict in the reference's key layout (hifigan_generator.py:199-234)."""
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    c0 = cfg["upsample_initial_channel"]
    _conv(sd, prefix + "conv_pre", c0, in_channels, 7, gen, wn=weight_norm and pre_wn)
    ch = c0
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(cfg["upsample_factors"], cfg["upsample_kernel_sizes"])):
        _conv(sd, prefix + "ups.%d" % i, ch // 2, ch, k, gen, wn=weight_norm, transposed=True,
              std=1.0 / math.sqrt(ch * k / u))
        ch //= 2
        for j, (rk, rd) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            rp = prefix + "resblocks.%d." % (i * nk + j)
            for m in range(len(rd)):
                if str(cfg["resblock_type"]) == "1":
                    _conv(sd, rp + "convs1.%d" % m, ch, ch, rk, gen, wn=weight_norm, std=0.7 / math.sqrt(ch * rk))
                    _conv(sd, rp + "convs2.%d" % m, ch, ch, rk, gen, wn=weight_norm, std=0.7 / math.sqrt(ch * rk))
                else:
                    _conv(sd, rp + "convs.%d" % m, ch, ch, rk, gen, wn=weight_norm, std=0.7 / math.sqrt(ch * rk))
    _conv(sd, prefix + "conv_post", out_channels, ch, 7, gen, wn=weight_norm and post_wn, bias=post_bias)
    return sd


class _F:
    def __init__(self, seed):
        self.gen = torch.Generator().manual_seed(seed)
        self.sd = {}

    def randn(self, *shape, std=1.0):
        return torch.randn(*shape, generator=self.gen) * std

    def conv(self, name, cout, cin, k, wn=False, bias=True, std=None, gain=1.0):
        std = std if std is not None else gain / math.sqrt(cin * k)
        v = self.randn(cout, cin, k, std=std)
        if wn:
            g = v.reshape(cout, -1).norm(dim=1).reshape(cout, 1, 1)
            g = g * (0.8 + 0.4 * torch.rand(cout, 1, 1, generator=self.gen))
            self.sd[name + ".parametrizations.weight.original0"] = g
            self.sd[name + ".parametrizations.weight.original1"] = v
        else:
            self.sd[name + ".weight"] = v
        if bias:
            self.sd[name + ".bias"] = self.randn(cout, std=0.05)

    def norm(self, name, c, shape3=False, gamma=1.0):
        shp = (1, c, 1) if shape3 else (c,)
        self.sd[name + ".gamma"] = gamma * (1.0 + 0.1 * self.randn(*shp))
        self.sd[name + ".beta"] = 0.05 * self.randn(*shp)


def _transformer(f, p, hidden, ffn_ch, layers, heads, k, window, ln3, gamma=1.0, out_channels=None):
    out_channels = out_channels or hidden
    for i in range(layers):
        a = p + "attn_layers.%d." % i
        if window is not None:
            f.sd[a + "emb_rel_k"] = f.randn(1, 2 * window + 1, hidden // heads, std=(hidden // heads) ** -0.5)
            f.sd[a + "emb_rel_v"] = f.randn(1, 2 * window + 1, hidden // heads, std=(hidden // heads) ** -0.5)
        for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
            f.conv(a + n, hidden, hidden, 1)
        f.norm(p + "norm_layers_1.%d" % i, hidden, ln3, gamma)
        last = (i + 1) == layers
        f.conv(p + "ffn_layers.%d.conv_1" % i, ffn_ch, hidden, k)
        f.conv(p + "ffn_layers.%d.conv_2" % i, out_channels if last else hidden, ffn_ch, k)
        f.norm(p + "norm_layers_2.%d" % i, out_channels if last else hidden, ln3, gamma)
    if out_channels != hidden:
        f.conv(p + "proj", out_channels, hidden, 1)


def _dds(f, p, c, k, layers):
    for i in range(layers):
        f.sd[p + "convs_sep.%d.weight" % i] = f.randn(c, 1, k, std=1 / math.sqrt(k))
        f.sd[p + "convs_sep.%d.bias" % i] = f.randn(c, std=0.05)
    for i in range(layers):
        f.conv(p + "convs_1x1.%d" % i, c, c, 1)
    for i in range(layers):
        f.norm(p + "norms_1.%d" % i, c)
    for i in range(layers):
        f.norm(p + "norms_2.%d" % i, c)


def _flows(f, p, hidden, k, nflows):
    f.sd[p + "0.translation"] = 0.1 * f.randn(2, 1)
    f.sd[p + "0.log_scale"] = 0.1 * f.randn(2, 1)
    for i in range(1, nflows + 1):
        q = p + "%d." % i
        f.conv(q + "pre", hidden, 1, 1)
        _dds(f, q + "convs.", hidden, k, 3)
        f.conv(q + "proj", 29, hidden, 1, std=0.5 / math.sqrt(hidden))  # zero-init in the reference


def _wn(f, p, hidden, k, layers, in_channels=None, cond=0):
    in_channels = in_channels or hidden
    if cond:  # wavenet.py:60-62: weight-normed 1x1 conv producing every layer's conditioning at once
        f.conv(p + "cond_layer", 2 * hidden * layers, cond, 1, wn=True, gain=0.5)
    for i in range(layers):
        f.conv(p + "in_layers.%d" % i, 2 * hidden, in_channels if i == 0 else hidden, k, wn=True)
        f.conv(p + "res_skip_layers.%d" % i, 2 * hidden if i < layers - 1 else hidden, hidden, 1, wn=True, gain=0.5)


def make_vits_state(args=None, seed=1234, with_decoder=True, with_posterior=False):
    """Reference-layout state_dict for the inference-relevant sub-modules of `Vits`
    (text_encoder., duration_predictor., flow., waveform_decoder.; vits.py:653-718)."""
    a = dict(VITS_DEFAULTS)
    a.update(args or {})
    h = a["hidden_channels"]
    f = _F(seed)
    p = "text_encoder."
    f.sd[p + "emb.weight"] = f.randn(a["num_chars"], h, std=h ** -0.5)
    # multilingual models: the language embedding rides as extra encoder channels (networks.py:62-63, vits.py:783-803)
    lng = int(a.get("embedded_language_dim", 4)) if a.get("use_language_embedding", False) else 0
    he = h + lng
    _transformer(f, p + "encoder.", he, a["hidden_channels_ffn_text_encoder"], a["num_layers_text_encoder"],
                 a["num_heads_text_encoder"], a["kernel_size_text_encoder"], 4, False)
    f.conv(p + "proj", 2 * h, he, 1, gain=0.5)
    if lng:
        f.sd["emb_l.weight"] = f.randn(int(a.get("num_languages", 3)), lng, std=1.0)
    spk = int(a.get("embedded_speaker_dim", 0) or 0)   # speaker_embedding_channels or d_vector_dim (vits.py:729-778)
    if spk and a.get("use_speaker_embedding", False):
        f.sd["emb_g.weight"] = f.randn(int(a.get("num_speakers", 4)), spk, std=0.5)
    p = "duration_predictor."
    if spk:
        f.conv(p + "cond", 192 if a["use_sdp"] else he, spk, 1, gain=0.5)
    if lng:
        f.conv(p + "cond_lang", 192 if a["use_sdp"] else he, lng, 1, gain=0.5)
    if a["use_sdp"]:
        f.conv(p + "pre", 192, he, 1)
        _dds(f, p + "convs.", 192, 3, 3)
        f.conv(p + "proj", 192, 192, 1)
        _flows(f, p + "flows.", 192, 3, 4)
        f.conv(p + "post_pre", 192, 1, 1)
        _dds(f, p + "post_convs.", 192, 3, 3)
        f.conv(p + "post_proj", 192, 192, 1)
        _flows(f, p + "post_flows.", 192, 3, 4)
    else:
        f.conv(p + "conv_1", 256, he, 3)
        f.norm(p + "norm_1", 256, True, 0.1)
        f.conv(p + "conv_2", 256, 256, 3)
        f.norm(p + "norm_2", 256, True, 0.1)
        f.conv(p + "proj", 1, 256, 1)
    for i in range(4):
        q = "flow.flows.%d." % i
        f.conv(q + "pre", h, h // 2, 1)
        _wn(f, q + "enc.", h, a["kernel_size_flow"], a["num_layers_flow"], cond=spk)
        f.conv(q + "post", h // 2, h, 1, gain=0.5)  # zero-init in the reference
    if with_posterior:  # PosteriorEncoder (networks.py:235-288), only used by voice conversion at inference time
        q = "posterior_encoder."
        f.conv(q + "pre", h, a.get("out_channels", 513), 1)
        _wn(f, q + "enc.", h, a.get("kernel_size_posterior_encoder", 5), a.get("num_layers_posterior_encoder", 16), cond=spk)
        f.conv(q + "proj", 2 * h, h, 1, gain=0.3)
    sd = f.sd
    if with_decoder:
        cfg = dict(resblock_type=a["resblock_type_decoder"], resblock_dilation_sizes=a["resblock_dilation_sizes_decoder"],
                   resblock_kernel_sizes=a["resblock_kernel_sizes_decoder"],
                   upsample_kernel_sizes=a["upsample_kernel_sizes_decoder"],
                   upsample_initial_channel=a["upsample_initial_channel_decoder"],
                   upsample_factors=a["upsample_rates_decoder"])
        sd.update(make_hifigan_state(cfg, h, seed=seed + 1, prefix="waveform_decoder.", pre_wn=False, post_wn=False,
                                     post_bias=False))
        if spk:  # hifigan_generator.py:215-216
            g2 = torch.Generator().manual_seed(seed + 2)
            sd["waveform_decoder.cond_layer.weight"] = torch.randn(a["upsample_initial_channel_decoder"], spk, 1,
                                                                   generator=g2) * (0.5 / math.sqrt(spk))
            sd["waveform_decoder.cond_layer.bias"] = torch.randn(a["upsample_initial_channel_decoder"], generator=g2) * 0.05
    return sd


def make_glow_state(args=None, seed=4321):
    """Reference-layout state_dict for `GlowTTS` (encoder., decoder.; glow_tts.py:80-105) after
    `store_inverse()` is NOT applied (weight-norm still parametrised, like a training checkpoint)."""
    a = dict(GLOW_DEFAULTS)
    a.update(args or {})
    ep = a["encoder_params"]
    h = a["hidden_channels_enc"]
    f = _F(seed)
    p = "encoder."
    f.sd[p + "emb.weight"] = f.randn(a["num_chars"], h, std=h ** -0.5)
    if a["use_encoder_prenet"]:
        for i in range(3):
            f.conv(p + "prenet.conv_layers.%d" % i, h, h, 5)
            f.norm(p + "prenet.norm_layers.%d" % i, h, True, 0.5)
        f.conv(p + "prenet.proj", h, h, 1, gain=0.5)  # zero-init in the reference
    ln3 = ep.get("layer_norm_type", "1") == "1"
    _transformer(f, p + "encoder.", h, ep["hidden_channels_ffn"], ep["num_layers"], ep["num_heads"], ep["kernel_size"],
                 ep.get("rel_attn_window_size"), ln3, gamma=0.5 if ln3 else 1.0)
    f.conv(p + "proj_m", a["out_channels"], h, 1)
    if not a["mean_only"]:
        f.conv(p + "proj_s", a["out_channels"], h, 1, gain=0.3)
    cin = int(a.get("c_in_channels", 0) or 0)      # speaker conditioning width (glow_tts.py:107-135)
    if cin and a.get("use_speaker_embedding", False) and not a.get("use_d_vector_file", False):
        f.sd["emb_g.weight"] = (torch.rand(int(a.get("num_speakers", 4)), h, generator=f.gen) - 0.5) * 0.2   # uniform(-0.1, 0.1)
    q = p + "duration_predictor."
    f.conv(q + "conv_1", a["hidden_channels_dp"], h + cin, 3)
    f.norm(q + "norm_1", a["hidden_channels_dp"], True, 0.5)
    f.conv(q + "conv_2", a["hidden_channels_dp"], a["hidden_channels_dp"], 3)
    f.norm(q + "norm_2", a["hidden_channels_dp"], True, 0.5)
    f.conv(q + "proj", 1, a["hidden_channels_dp"], 1)
    c = a["out_channels"] * a["num_squeeze"]
    hd = a["hidden_channels_dec"]
    for b in range(a["num_flow_blocks_dec"]):
        f.sd["decoder.flows.%d.logs" % (3 * b)] = 0.1 * f.randn(1, c, 1)
        f.sd["decoder.flows.%d.bias" % (3 * b)] = 0.1 * f.randn(1, c, 1)
        w = torch.linalg.qr(f.randn(a["num_splits"], a["num_splits"]), "complete")[0]
        f.sd["decoder.flows.%d.weight" % (3 * b + 1)] = w + 0.05 * f.randn(a["num_splits"], a["num_splits"])
        q = "decoder.flows.%d." % (3 * b + 2)
        f.conv(q + "start", hd, c // 2, 1, wn=True)
        f.conv(q + "end", c, hd, 1, gain=0.3)  # zero-init in the reference
        _wn(f, q + "wn.", hd, a["kernel_size_dec"], a["num_block_layers"], cond=cin)
    return f.sd


HIFIGAN_V1 = dict(  # TTS/vocoder/configs/hifigan_config.py:95-104
    resblock_type="1", resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], resblock_kernel_sizes=[3, 7, 11],
    upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=512, upsample_factors=[8, 8, 2, 2],
    inference_padding=5)
HIFIGAN_V2 = dict(HIFIGAN_V1, upsample_initial_channel=128)  # HiFi-GAN paper V2


def make_hifi_decoder_state(decoder_input_dim=1024, d_vector_dim=512, upsample_initial_channel=512, seed=77):
    """`waveform_decoder.*` keys of the XTTS HifiDecoder (TTS/tts/layers/xtts/hifigan_decoder.py:615-667): a HiFiGAN-v1
    generator without weight-norm on conv_pre/conv_post, no conv_post bias, `cond_layer` + one `conds.i` per upsample."""
    cfg = dict(HIFIGAN_V1, upsample_initial_channel=upsample_initial_channel, inference_padding=0)
    sd = make_hifigan_state(cfg, decoder_input_dim, seed=seed, prefix="waveform_decoder.", pre_wn=False, post_wn=False,
                            post_bias=False)
    g = torch.Generator().manual_seed(seed + 1)
    std = 0.5 / math.sqrt(d_vector_dim)
    sd["waveform_decoder.cond_layer.weight"] = torch.randn(upsample_initial_channel, d_vector_dim, 1, generator=g) * std
    sd["waveform_decoder.cond_layer.bias"] = torch.randn(upsample_initial_channel, generator=g) * 0.05
    ch = upsample_initial_channel
    for i in range(len(cfg["upsample_factors"])):
        ch //= 2
        sd["waveform_decoder.conds.%d.weight" % i] = torch.randn(ch, d_vector_dim, 1, generator=g) * std
        sd["waveform_decoder.conds.%d.bias" % i] = torch.randn(ch, generator=g) * 0.05
    return sd, cfg
