"""Golden cases: each function runs either the REAL reference modules (`impl="ref"`, container only)
or the oracle restatement (`impl="oracle"`) on identical seeded weights + inputs and returns the
tensors that are compared."""
import torch

from oracle import tts_oracle as O
from oracle import weights as W


def _g(seed):
    return torch.Generator().manual_seed(seed)


def hifigan_small(impl, rb="1"):
    cfg = dict(W.HIFIGAN_V1, upsample_initial_channel=64, resblock_type=rb)
    if rb == "2":
        cfg["resblock_dilation_sizes"] = [[1, 3], [1, 3], [1, 3]]
    sd = O.make_hifigan_state(cfg, 80, seed=11)
    x = torch.randn(2, 80, 23, generator=_g(0))
    if impl == "ref":
        from oracle import ref_models as RM

        with torch.no_grad():
            o = RM.hifigan(sd, cfg, 80).inference(x)
    else:
        o = O.hifigan_inference(sd, "", x, cfg)
    return {"wav": o}


HIFIGAN_UNTUNED = dict(W.HIFIGAN_V1, upsample_initial_channel=64, resblock_kernel_sizes=[5, 9, 3],
                       resblock_dilation_sizes=[[1, 2, 4], [2, 6, 3], [3, 12, 1]], upsample_factors=[3, 2, 4, 2], upsample_kernel_sizes=[7, 4, 4, 6])


def hifigan_untuned(impl):
    """A generator outside every released config (hifigan_generator.py:199-233 accepts any): ResBlock kernels 5 / 9 / 3 at the dilations
    (1,2,4) (2,6,3) (3,12,1) (ResBlock1 takes exactly three: hifigan_generator.py:36-66), upsampling (stride, kernel) = (3,7) (2,4) (4,4) (2,6) — the padding rules `(k - u) // 2` / `get_padding` at
    sizes the default configs never reach.  The GPU path runs this config on the generic conv / polyphase fallback
    (tests/test_hifigan_gpu.py) against the oracle; this case pins the oracle itself to the reference module."""
    cfg = dict(HIFIGAN_UNTUNED)
    sd = O.make_hifigan_state(cfg, 80, seed=29)
    x = torch.randn(2, 80, 33, generator=_g(2))
    if impl == "ref":
        from oracle import ref_models as RM

        with torch.no_grad():
            o = RM.hifigan(sd, cfg, 80).inference(x)
    else:
        o = O.hifigan_inference(sd, "", x, cfg)
    return {"wav": o}


HIFIGAN_TRAINED_LIKE = dict(W.HIFIGAN_V1, upsample_initial_channel=256)


def trained_like_hifigan_state(cfg, seed):
    """Stand-in for a released checkpoint (unreachable offline): seeded weights re-scaled the way training leaves them — weight-norm
    gains `g` spread over 1e-2 .. 1e1 per output channel in every ResBlock's first conv, the following conv's input columns carrying
    the inverse factor (leaky ReLU is positively homogeneous, so the network function is unchanged while the intermediate tensors
    span three decades per tile and the second conv's weight rows mix magnitudes 1e-1 .. 1e2), and mel channels spanning
    1e-5 .. 1e2 with conv_pre's columns compensating.  Returns (state_dict, per-mel-channel input scale)."""
    sd = O.make_hifigan_state(cfg, 80, seed=seed)
    gen = _g(seed + 1)
    g0, v1 = "parametrizations.weight.original0", "parametrizations.weight.original1"
    nk = len(cfg["resblock_kernel_sizes"])
    for blk in range(len(cfg["upsample_factors"]) * nk):
        for m in range(3):
            a, b = "resblocks.%d.convs1.%d." % (blk, m), "resblocks.%d.convs2.%d." % (blk, m)
            c = sd[a + g0].shape[0]
            s = 10.0 ** (torch.rand(c, generator=gen) * 3.0 - 2.0)                 # 1e-2 .. 1e1 per channel
            sd[a + g0] = sd[a + g0] * s.view(c, 1, 1)
            sd[a + "bias"] = sd[a + "bias"] * s
            v = sd[b + v1]
            ratio = sd[b + g0] / v.reshape(v.shape[0], -1).norm(dim=1).view(-1, 1, 1)
            v = v / s.view(1, c, 1)
            sd[b + v1] = v
            sd[b + g0] = ratio * v.reshape(v.shape[0], -1).norm(dim=1).view(-1, 1, 1)  # same effective weight, columns / s
    sx = 10.0 ** (torch.rand(80, generator=gen) * 7.0 - 5.0)                        # mel channels: 1e-5 .. 1e2
    v = sd["conv_pre." + v1]
    ratio = sd["conv_pre." + g0] / v.reshape(v.shape[0], -1).norm(dim=1).view(-1, 1, 1)
    v = v / sx.view(1, 80, 1)
    sd["conv_pre." + v1] = v
    sd["conv_pre." + g0] = ratio * v.reshape(v.shape[0], -1).norm(dim=1).view(-1, 1, 1)
    return sd, sx


def hifigan_trained_like(impl):
    """See trained_like_hifigan_state; 2 x 130 frames so that the 128-, 64- and 32-channel stages run on the large-grid
    (three-product) kernels of the GPU path (tests/test_hifigan_gpu.py)."""
    cfg = dict(HIFIGAN_TRAINED_LIKE)
    sd, sx = trained_like_hifigan_state(cfg, 77)
    x = torch.randn(2, 80, 130, generator=_g(5)) * sx.view(1, 80, 1)
    if impl == "ref":
        from oracle import ref_models as RM

        with torch.no_grad():
            o = RM.hifigan(sd, cfg, 80).inference(x)
    else:
        o = O.hifigan_inference(sd, "", x, cfg)
    return {"wav": o}


VITS_SMALL = dict(upsample_initial_channel_decoder=64)


def vits_small(impl, use_sdp=True):
    args = dict(VITS_SMALL, use_sdp=use_sdp)
    sd = W.make_vits_state(args, seed=1234)
    x = torch.randint(0, 100, (3, 37), generator=_g(0))
    xl = torch.tensor([37, 30, 21])
    if impl == "ref":
        from oracle import ref_models as RM

        out = RM.RefVits(sd, args).inference(x, xl, seed=7)
    else:
        torch.manual_seed(7)
        out = O.vits_inference(sd, x, xl, args)
    keys = ["x", "logw", "durations", "m_p", "logs_p", "z_p", "z", "model_outputs"]
    return {k: out[k] for k in keys}


def trained_like_vits_state(args, seed):
    """Stand-in for a released VITS checkpoint (unreachable offline): the seeded weights of make_vits_state re-scaled the way
    training leaves them, so that the acoustic half sees what the random initialisation never produces:
      * flow WaveNets: weight-norm gains of every gate conv (in_layers) spread over 1e-2 .. 1e1 per output row — pre-activations
        from the linear region of tanh / sigmoid to +-10 and beyond (saturated gates); the skip rows of every res/skip conv carry a
        per-channel factor 1e-2 .. 1e1 that `post`'s input columns undo (the coupling's function is unchanged, the skip
        accumulator spans three decades per tile);
      * LayerNorm gains of the text encoder and of every DDSConv spread over 1e-1 .. 10^0.5 per channel (sharp attention, wide-range
        FFN / depth-separable activations), the text encoder's projection columns compensating its last norm;
      * ConvFlow spline parameters: the width / height rows of every `proj` scaled by 1 .. 200 per row — bin widths and heights
        collapse to the minimum (1e-3) next to dominant bins, so knots nearly coincide and inputs land next to bin edges;
      * the waveform decoder re-scaled as trained_like_hifigan_state does (ResBlock gains 1e-2 .. 1e1 with the next conv's columns
        compensating)."""
    sd = W.make_vits_state(args, seed=seed)
    gen = _g(seed + 7)
    g0, v1 = "parametrizations.weight.original0", "parametrizations.weight.original1"
    h = 192
    spread = lambda n, lo, hi: 10.0 ** (torch.rand(n, generator=gen) * (hi - lo) + lo)  # noqa: E731
    for i in range(4):
        q = "flow.flows.%d." % i
        s_skip = spread(h, -2.0, 1.0)
        for l in range(4):
            a = q + "enc.in_layers.%d." % l
            s = spread(2 * h, -2.0, 1.0)
            sd[a + g0] = sd[a + g0] * s.view(-1, 1, 1)
            sd[a + "bias"] = sd[a + "bias"] * s
            r = q + "enc.res_skip_layers.%d." % l
            rows = sd[r + g0].shape[0]
            f = torch.ones(rows)
            f[rows - h:] = s_skip                                   # skip rows: the second half (all rows of the last layer)
            sd[r + g0] = sd[r + g0] * f.view(-1, 1, 1)
            sd[r + "bias"] = sd[r + "bias"] * f
        sd[q + "post.weight"] = sd[q + "post.weight"] / s_skip.view(1, h, 1)
    for k in list(sd):
        if k.startswith("text_encoder.encoder.norm_layers_") and k.endswith(".gamma"):
            sd[k] = sd[k] * spread(sd[k].numel(), -1.0, 0.5).view_as(sd[k])
        if k.startswith("duration_predictor.") and ".norms_" in k and k.endswith(".gamma") and "post_" not in k:
            sd[k] = sd[k] * spread(sd[k].numel(), -1.0, 0.5).view_as(sd[k])
    last = "text_encoder.encoder.norm_layers_2.%d.gamma" % (5)
    sd["text_encoder.proj.weight"] = sd["text_encoder.proj.weight"] / sd[last].abs().clamp_min(1e-3).view(1, -1, 1) * 0.1
    if args.get("use_sdp", True):
        for i in range(1, 5):
            k = "duration_predictor.flows.%d.proj." % i
            rows = sd[k + "weight"].shape[0]                         # 10 widths | 10 heights | 9 derivatives
            f = torch.ones(rows)
            f[:20] = spread(20, 0.0, 2.3)
            sd[k + "weight"] = sd[k + "weight"] * f.view(-1, 1, 1)
            sd[k + "bias"] = sd[k + "bias"] * f
        # the last reverse flow (ElementwiseAffine) brings the spline's +-5 range back to durations of a few frames per token
        sd["duration_predictor.flows.0.log_scale"] = sd["duration_predictor.flows.0.log_scale"] + 1.5
        sd["duration_predictor.flows.0.translation"] = sd["duration_predictor.flows.0.translation"] - 2.0
    # waveform decoder (conv_pre / conv_post carry no weight norm inside VITS, vits.py:704-718)
    nk = 3
    for blk in range(4 * nk):
        for m in range(3):
            a, b = "waveform_decoder.resblocks.%d.convs1.%d." % (blk, m), "waveform_decoder.resblocks.%d.convs2.%d." % (blk, m)
            c = sd[a + g0].shape[0]
            s = spread(c, -2.0, 1.0)
            sd[a + g0] = sd[a + g0] * s.view(c, 1, 1)
            sd[a + "bias"] = sd[a + "bias"] * s
            v = sd[b + v1]
            ratio = sd[b + g0] / v.reshape(v.shape[0], -1).norm(dim=1).view(-1, 1, 1)
            v = v / s.view(1, c, 1)
            sd[b + v1] = v
            sd[b + g0] = ratio * v.reshape(v.shape[0], -1).norm(dim=1).view(-1, 1, 1)
    return sd


VITS_TRAINED_LIKE = dict(upsample_initial_channel_decoder=64)


def vits_trained_like(impl, use_sdp=True):
    """See trained_like_vits_state.  The GPU test (tests/test_vits_gpu.py) runs this fixture on the small-grid kernels a request of
    this size takes by itself AND with the large-grid tiles forced (ttsamd_conv1d_set_small_grid(0)): the three-product / six-product /
    fp32 GATE, RES_SKIP and COUPLE epilogues of the B = 32 step."""
    args = dict(VITS_TRAINED_LIKE, use_sdp=use_sdp)
    sd = trained_like_vits_state(args, 4321)
    x = torch.randint(0, 100, (3, 37), generator=_g(0))
    xl = torch.tensor([37, 30, 21])
    if impl == "ref":
        from oracle import ref_models as RM

        out = RM.RefVits(sd, args).inference(x, xl, seed=7)
    else:
        torch.manual_seed(7)
        out = O.vits_inference(sd, x, xl, args)
    keys = ["logw", "durations", "z_p", "z", "model_outputs"]
    res = {k: out[k] for k in keys}
    # The collapsed spline bins make logw ill-conditioned where an input lands next to a knot (slopes of ~1e2 .. 1e3): the fp32
    # reference itself is 4e-5 away from an fp64 evaluation there.  The fixture carries that fp64 witness (the oracle restatement —
    # bitwise the reference modules in fp32, tests/test_oracle_pin.py — run in double on the same weights, noise and inputs), so
    # that a GPU test can bound its error by the reference's own rounding error instead of by a fixed 1e-5.
    torch.manual_seed(7)
    noise_dp = torch.randn(3, 2, 37)
    o64 = O.vits_inference({k: v.double() for k, v in sd.items()}, x, xl, args, noise_dp=noise_dp.double(), stop_after="prior",
                           noise_z=torch.zeros(3, 192, 1, dtype=torch.float64))
    res["logw_fp64"] = o64["logw"]
    return res


def vits_small_speaker(impl, mode):
    """Multi-speaker conditioning (vits.py:873-886,1116-1117): speaker-embedding table or external d-vectors feeding the
    duration predictor, every flow WN and the waveform decoder."""
    args = dict(VITS_SMALL, embedded_speaker_dim=24, use_speaker_embedding=(mode == "emb"), num_speakers=5,
                use_sdp=(mode == "emb"))
    sd = W.make_vits_state(args, seed=4242)
    x = torch.randint(0, 100, (2, 25), generator=_g(3))
    xl = torch.tensor([25, 16])
    sid = torch.tensor([3, 1]) if mode == "emb" else None
    dv = None if mode == "emb" else torch.randn(2, 24, generator=_g(4))
    if impl == "ref":
        from oracle import ref_models as RM

        out = RM.RefVits(sd, args).inference(x, xl, seed=9, speaker_ids=sid, d_vectors=dv)
    else:
        torch.manual_seed(9)
        out = O.vits_inference(sd, x, xl, args, g=O.vits_speaker_g(sd, sid, dv))
    return {k: out[k] for k in ["logw", "durations", "z_p", "z", "model_outputs"]}


def vits_small_language(impl, use_sdp):
    """Multilingual VITS (vits.py:783-803,1119-1138; networks.py:62-63,89-91): the language embedding widens the text
    encoder to 192+4 channels (head size 98) and conditions the duration predictor next to the speaker embedding."""
    args = dict(VITS_SMALL, embedded_speaker_dim=24, use_speaker_embedding=True, num_speakers=5, use_sdp=use_sdp,
                use_language_embedding=True, embedded_language_dim=4, num_languages=3)
    sd = W.make_vits_state(args, seed=777)
    x = torch.randint(0, 100, (2, 23), generator=_g(13))
    xl = torch.tensor([23, 17])
    sid, lid = torch.tensor([2, 4]), torch.tensor([1, 2])
    if impl == "ref":
        from oracle import ref_models as RM

        out = RM.RefVits(sd, args).inference(x, xl, seed=19, speaker_ids=sid, language_ids=lid)
    else:
        torch.manual_seed(19)
        out = O.vits_inference(sd, x, xl, args, g=O.vits_speaker_g(sd, sid, None), lang_emb=O.vits_language_emb(sd, lid))
    return {k: out[k] for k in ["logw", "durations", "z_p", "z", "model_outputs"]}


VITS_VC = dict(VITS_SMALL, embedded_speaker_dim=24, use_speaker_embedding=True, num_speakers=5, out_channels=65,
               num_layers_posterior_encoder=6)


def vits_voice_conversion(impl):
    """Vits.voice_conversion (vits.py:1202-1228): posterior encoder + flow forward (source speaker) + flow reverse and
    decoder (target speaker)."""
    sd = W.make_vits_state(VITS_VC, seed=555, with_posterior=True)
    y = torch.randn(2, 65, 40, generator=_g(8))
    yl = torch.tensor([40, 27])
    g_src = torch.nn.functional.embedding(torch.tensor([0, 2]), sd["emb_g.weight"]).unsqueeze(-1)
    g_tgt = torch.nn.functional.embedding(torch.tensor([4, 1]), sd["emb_g.weight"]).unsqueeze(-1)
    if impl == "ref":
        from oracle import ref_models as RM

        out = RM.RefVits(sd, VITS_VC).voice_conversion(y, yl, g_src, g_tgt, seed=13)
    else:
        torch.manual_seed(13)
        out = O.vits_voice_conversion(sd, y, yl, g_src, g_tgt, VITS_VC)
    return {k: out[k] for k in ["z", "z_p", "z_hat", "model_outputs"]}


def xtts_hifi_decoder(impl):
    """XTTS HifiDecoder vocoder half (xtts/hifigan_decoder.py:675-701): interpolated GPT latents + d-vector conditioning
    at the input and after every upsampling layer."""
    sd, cfg = W.make_hifi_decoder_state(decoder_input_dim=96, d_vector_dim=32, upsample_initial_channel=64, seed=31)
    lat = torch.randn(2, 9, 96, generator=_g(5))
    g = torch.randn(2, 32, 1, generator=_g(6))
    if impl == "ref":
        from oracle import ref_shim

        m = ref_shim.ref("TTS.tts.layers.xtts.hifigan_decoder")
        net = m.HifiganGenerator(96, 1, "1", cfg["resblock_dilation_sizes"], cfg["resblock_kernel_sizes"],
                                 cfg["upsample_kernel_sizes"], 64, cfg["upsample_factors"], inference_padding=0,
                                 cond_channels=32, conv_pre_weight_norm=False, conv_post_weight_norm=False,
                                 conv_post_bias=False, cond_in_each_up_layer=True).eval()
        net.load_state_dict({k[len("waveform_decoder."):]: v for k, v in sd.items()}, strict=True)
        with torch.no_grad():   # HifiDecoder.forward, restated over the real generator (the class itself needs torchaudio)
            z = torch.nn.functional.interpolate(lat.transpose(1, 2), scale_factor=[1024 / 256], mode="linear").squeeze(1)
            z = torch.nn.functional.interpolate(z, scale_factor=[24000 / 22050], mode="linear").squeeze(0)
            o = net(z, g=g)
    else:
        o = O.hifi_decoder_forward(sd, lat, g, cfg)
    return {"wav": o}


GLOW_SMALL = dict(inference_noise_scale=0.33, num_flow_blocks_dec=4)


def glow_small(impl, window=None, ln="1"):
    args = dict(GLOW_SMALL)
    args["encoder_params"] = dict(O.GLOW_DEFAULTS["encoder_params"], rel_attn_window_size=window, layer_norm_type=ln,
                                  num_layers=3)
    sd = W.make_glow_state(args, seed=4321)
    x = torch.randint(0, 130, (2, 29), generator=_g(1))
    xl = torch.tensor([29, 20])
    if impl == "ref":
        from oracle import ref_models as RM

        out = RM.RefGlow(sd, args).inference(x, xl, seed=3)
    else:
        torch.manual_seed(3)
        out = O.glow_tts_inference(sd, x, xl, args)
    return {k: out[k] for k in ["model_outputs", "durations", "durations_log", "y_mean"]}


def glow_small_speaker(impl, mode):
    """Multi-speaker Glow-TTS (glow_tts.py:107-135,179-191; encoder.py:166-168; glow.py:199-213): the normalised
    speaker vector is concatenated to the duration predictor's input and conditions every coupling WaveNet."""
    cin = 192 if mode == "emb" else 48
    args = dict(GLOW_SMALL, c_in_channels=cin, use_speaker_embedding=(mode == "emb"), use_d_vector_file=(mode == "dvec"),
                num_speakers=4, d_vector_dim=48)
    args["encoder_params"] = dict(O.GLOW_DEFAULTS["encoder_params"], num_layers=2)
    sd = W.make_glow_state(args, seed=909)
    x = torch.randint(0, 130, (2, 21), generator=_g(21))
    xl = torch.tensor([21, 13])
    sid = torch.tensor([3, 0]) if mode == "emb" else None
    dv = None if mode == "emb" else torch.randn(2, 48, generator=_g(22))
    if impl == "ref":
        from oracle import ref_models as RM

        out = RM.RefGlow(sd, args).inference(x, xl, seed=5, speaker_ids=sid, d_vectors=dv)
    else:
        torch.manual_seed(5)
        out = O.glow_tts_inference(sd, x, xl, args, g=O.glow_speaker_g(sd, sid, dv))
    return {k: out[k] for k in ["model_outputs", "durations", "durations_log", "y_mean"]}


CASES = {
    "hifigan_small_rb1": lambda impl: hifigan_small(impl, "1"),
    "hifigan_small_rb2": lambda impl: hifigan_small(impl, "2"),
    "hifigan_untuned": hifigan_untuned,
    "hifigan_trained_like": hifigan_trained_like,
    "vits_small_sdp": lambda impl: vits_small(impl, True),
    "vits_small_dp": lambda impl: vits_small(impl, False),
    "vits_trained_like": vits_trained_like,
    "vits_small_spk_emb": lambda impl: vits_small_speaker(impl, "emb"),
    "vits_small_spk_dvec": lambda impl: vits_small_speaker(impl, "dvec"),
    "vits_small_lang_sdp": lambda impl: vits_small_language(impl, True),
    "vits_small_lang_dp": lambda impl: vits_small_language(impl, False),
    "xtts_hifi_decoder": xtts_hifi_decoder,
    "vits_voice_conversion": vits_voice_conversion,
    "glow_small": lambda impl: glow_small(impl),
    "glow_small_spk_emb": lambda impl: glow_small_speaker(impl, "emb"),
    "glow_small_spk_dvec": lambda impl: glow_small_speaker(impl, "dvec"),
    "glow_small_relwin": lambda impl: glow_small(impl, 4, "2"),
}
