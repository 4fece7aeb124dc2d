"""TEST INFRASTRUCTURE ONLY — CPU fp32 restatement (plain torch ops) of the reference's VITS /
Glow-TTS / HiFiGAN inference path, driven by a reference-layout `state_dict`.

This is the parity checker for the floating-point HIP kernels (tolerances are written in the
tests).  It is a *restatement*: every function cites the reference file:line it follows and is
pinned against the reference's own `nn.Module`s (imported through oracle/ref_shim.py in the build
container; see tests/test_oracle_pin.py and the committed fixtures under tests/golden/ made by
tests/golden/make_golden.py).  It runs without /root/reference (GPU box).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product
package `tts_amd` never does.

Conventions: `sd` is a flat dict name -> tensor with the reference's parameter names
(weight-norm'ed convs appear as `<name>.parametrizations.weight.original0/1`, legacy
`weight_g/weight_v`, or already-folded `<name>.weight`), `p` is the key prefix ending in '.'.
All tensors fp32, channels-first [B, C, T].
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # hifigan_generator.py:11


# ----------------------------------------------------------------------------------------------
# parameter access
# ----------------------------------------------------------------------------------------------
def weight(sd, name):
    """Effective conv weight. torch weight_norm: w = v * (g / ||v||), norm over all dims but 0
    (`torch._weight_norm(v, g, 0)`) — for ConvTranspose1d dim 0 is in_channels."""
    if name + ".weight" in sd:
        return sd[name + ".weight"]
    g = sd.get(name + ".parametrizations.weight.original0", sd.get(name + ".weight_g"))
    v = sd.get(name + ".parametrizations.weight.original1", sd.get(name + ".weight_v"))
    if g is None or v is None:
        raise KeyError(name)
    return torch._weight_norm(v, g, 0)


def bias(sd, name):
    return sd.get(name + ".bias")


def conv1d(sd, name, x, dilation=1, padding=0, groups=1):
    return F.conv1d(x, weight(sd, name), bias(sd, name), 1, padding, dilation, groups)


def sequence_mask(lengths, max_len=None):
    """TTS/tts/utils/helpers.py:43-57."""
    if max_len is None:
        max_len = int(lengths.max())
    r = torch.arange(max_len, dtype=lengths.dtype, device=lengths.device)
    return r.unsqueeze(0) < lengths.unsqueeze(1)


def generate_path(duration, mask):
    """TTS/tts/utils/helpers.py:154-169."""
    b, t_x, t_y = mask.shape
    cum = torch.cumsum(duration, 1)
    path = sequence_mask(cum.view(b * t_x), t_y).to(mask.dtype).view(b, t_x, t_y)
    path = path - F.pad(path, [0, 0, 1, 0, 0, 0])[:, :-1]
    return path * mask


# ----------------------------------------------------------------------------------------------
# HiFiGAN generator — TTS/vocoder/models/hifigan_generator.py
# ----------------------------------------------------------------------------------------------
def get_padding(k, d):
    return int((k * d - d) / 2)  # hifigan_generator.py:14-15


def resblock1(sd, p, x, k, dil):
    """ResBlock1.forward, hifigan_generator.py:83-98."""
    for i, d in enumerate(dil):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = conv1d(sd, p + "convs1.%d" % i, xt, dilation=d, padding=get_padding(k, d))
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = conv1d(sd, p + "convs2.%d" % i, xt, dilation=1, padding=get_padding(k, 1))
        x = xt + x
    return x


def resblock2(sd, p, x, k, dil):
    """ResBlock2.forward, hifigan_generator.py:150-155."""
    for i, d in enumerate(dil):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = conv1d(sd, p + "convs.%d" % i, xt, dilation=d, padding=get_padding(k, d))
        x = xt + x
    return x


def hifigan_forward(sd, p, x, cfg, g=None):
    """HifiganGenerator.forward, hifigan_generator.py:236-265.

    cfg keys (constructor args, hifigan_generator.py:163-178): resblock_type, resblock_dilation_sizes,
    resblock_kernel_sizes, upsample_kernel_sizes, upsample_factors (upsample_initial_channel is
    implied by the weights)."""
    o = conv1d(sd, p + "conv_pre", x, padding=3)
    if g is not None and (p + "cond_layer.weight") in sd:
        o = o + conv1d(sd, p + "cond_layer", g)
    nk = len(cfg["resblock_kernel_sizes"])
    rb = resblock1 if str(cfg["resblock_type"]) == "1" else resblock2
    for i, (u, k) in enumerate(zip(cfg["upsample_factors"], cfg["upsample_kernel_sizes"])):
        o = F.leaky_relu(o, LRELU_SLOPE)
        o = F.conv_transpose1d(o, weight(sd, p + "ups.%d" % i), bias(sd, p + "ups.%d" % i), u, (k - u) // 2)
        if g is not None and (p + "conds.%d.weight" % i) in sd:   # XTTS variant, xtts/hifigan_decoder.py:283-284
            o = o + conv1d(sd, p + "conds.%d" % i, g)
        z_sum = None
        for j in range(nk):
            r = rb(sd, p + "resblocks.%d." % (i * nk + j), o, cfg["resblock_kernel_sizes"][j],
                   cfg["resblock_dilation_sizes"][j])
            z_sum = r if z_sum is None else z_sum + r
        o = z_sum / nk
    o = F.leaky_relu(o)  # default slope 0.01 (hifigan_generator.py:262)
    o = conv1d(sd, p + "conv_post", o, padding=3)
    return torch.tanh(o)


def hifi_decoder_forward(sd, latents, g, cfg, input_sample_rate=22050, output_sample_rate=24000, output_hop_length=256,
                         ar_mel_length_compression=1024):
    """HifiDecoder.forward, TTS/tts/layers/xtts/hifigan_decoder.py:675-701: latents [B,T,C] -> two linear
    interpolations -> the conditioned generator (prefix `waveform_decoder.`)."""
    z = F.interpolate(latents.transpose(1, 2), scale_factor=[ar_mel_length_compression / output_hop_length],
                      mode="linear").squeeze(1)
    if output_sample_rate != input_sample_rate:
        z = F.interpolate(z, scale_factor=[output_sample_rate / input_sample_rate], mode="linear").squeeze(0)
    return hifigan_forward(sd, "waveform_decoder.", z, cfg, g=g)


def xtts_handle_chunks(wav_gen, wav_gen_prev, wav_overlap, overlap_len):
    """Xtts.handle_chunks, TTS/tts/models/xtts.py:585-607 (pinned against the function body itself, extracted from the
    reference file, in tests/test_oracle_pin.py)."""
    lo = 0 if wav_gen_prev is None else wav_gen_prev.shape[0] - overlap_len
    wav_chunk = wav_gen[lo:-overlap_len]
    if wav_overlap is not None:
        if overlap_len > len(wav_chunk):
            return (wav_gen[lo:] if wav_gen_prev is not None else wav_gen[-overlap_len:]), wav_gen, None
        up = torch.linspace(0.0, 1.0, overlap_len)
        down = torch.linspace(1.0, 0.0, overlap_len)
        mixed = wav_chunk[:overlap_len] * up
        wav_chunk[:overlap_len] = wav_overlap * down
        wav_chunk[:overlap_len] += mixed
    return wav_chunk, wav_gen, wav_gen[-overlap_len:]


def xtts_stream_decode(sd, latent_steps, g, cfg, stream_chunk_size=20, overlap_wav_len=1024, length_scale=1.0, **dec):
    """Vocoder half of Xtts.inference_stream, xtts.py:653-687: every `stream_chunk_size` latents (and at the end) the
    whole prefix is re-vocoded and `handle_chunks` cuts + cross-fades.  latent_steps: list of [C] tensors."""
    chunks, seen, fresh = [], [], 0
    prev, overlap = None, None
    steps = list(latent_steps)
    i, end = 0, False
    while not end:
        if i < len(steps):
            seen.append(steps[i].reshape(1, -1))
            fresh += 1
            i += 1
        else:
            end = True
        if end or (stream_chunk_size > 0 and fresh >= stream_chunk_size):
            lat = torch.cat(seen, 0)[None]
            if length_scale != 1.0:
                lat = F.interpolate(lat.transpose(1, 2), scale_factor=length_scale, mode="linear").transpose(1, 2)
            wav = hifi_decoder_forward(sd, lat, g, cfg, **dec)
            chunk, prev, overlap = xtts_handle_chunks(wav.squeeze(), prev, overlap, overlap_wav_len)
            fresh = 0
            chunks.append(chunk)
    return chunks


def hifigan_inference(sd, p, c, cfg):
    """HifiganGenerator.inference, hifigan_generator.py:267-282 (replicate pad, no crop)."""
    pad = cfg.get("inference_padding", 5)
    c = F.pad(c, (pad, pad), "replicate")
    return hifigan_forward(sd, p, c, cfg)


# ----------------------------------------------------------------------------------------------
# WaveNet block / coupling flows — generic/wavenet.py, vits/networks.py
# ----------------------------------------------------------------------------------------------
def wn_forward(sd, p, x, x_mask, hidden, kernel_size, dilation_rate, num_layers, g=None):
    """WN.forward, TTS/tts/layers/generic/wavenet.py:92-116 (+ fused gate :6-13)."""
    output = torch.zeros_like(x)
    if g is not None:
        g = conv1d(sd, p + "cond_layer", g)
    for i in range(num_layers):
        d = dilation_rate ** i
        x_in = conv1d(sd, p + "in_layers.%d" % i, x, dilation=d, padding=int((kernel_size * d - d) / 2))
        if g is not None:
            g_l = g[:, i * 2 * hidden:(i + 1) * 2 * hidden, :]
        else:
            g_l = torch.zeros_like(x_in)
        in_act = x_in + g_l
        acts = torch.tanh(in_act[:, :hidden]) * torch.sigmoid(in_act[:, hidden:])
        rs = conv1d(sd, p + "res_skip_layers.%d" % i, acts)
        if i < num_layers - 1:
            x = (x + rs[:, :hidden]) * x_mask
            output = output + rs[:, hidden:]
        else:
            output = output + rs
    return output * x_mask


def residual_coupling_reverse(sd, p, x, x_mask, cfg, g=None):
    """ResidualCouplingBlock.forward(reverse=True), vits/networks.py:138-166 (mean_only=True)."""
    half = x.shape[1] // 2
    x0, x1 = x[:, :half], x[:, half:]
    h = conv1d(sd, p + "pre", x0) * x_mask
    h = wn_forward(sd, p + "enc.", h, x_mask, cfg["hidden"], cfg["kernel_size"], cfg["dilation_rate"],
                   cfg["num_layers"], g=g)
    stats = conv1d(sd, p + "post", h) * x_mask
    if stats.shape[1] == half:  # mean_only
        m, log_scale = stats, torch.zeros_like(stats)
    else:
        m, log_scale = stats[:, :half], stats[:, half:]
    x1 = (x1 - m) * torch.exp(-log_scale) * x_mask
    return torch.cat([x0, x1], 1)


def residual_coupling_forward(sd, p, x, x_mask, cfg, g=None):
    """ResidualCouplingBlock.forward(reverse=False), vits/networks.py:147-161 (mean_only=True)."""
    half = x.shape[1] // 2
    x0, x1 = x[:, :half], x[:, half:]
    h = conv1d(sd, p + "pre", x0) * x_mask
    h = wn_forward(sd, p + "enc.", h, x_mask, cfg["hidden"], cfg["kernel_size"], cfg["dilation_rate"],
                   cfg["num_layers"], g=g)
    m = conv1d(sd, p + "post", h) * x_mask
    x1 = m + x1 * torch.exp(torch.zeros_like(m)) * x_mask
    return torch.cat([x0, x1], 1)


def residual_coupling_blocks_forward(sd, p, x, x_mask, cfg, g=None):
    """ResidualCouplingBlocks.forward(reverse=False), vits/networks.py:221-225."""
    for i in range(cfg.get("num_flows", 4)):
        x = residual_coupling_forward(sd, p + "flows.%d." % i, x, x_mask, cfg, g=g)
        x = torch.flip(x, [1])
    return x


def posterior_encoder(sd, p, y, y_lengths, hidden, kernel_size, dilation_rate, num_layers, g=None, noise=None):
    """PosteriorEncoder.forward, vits/networks.py:275-288.  `noise` replaces randn_like(mean)."""
    x_mask = torch.unsqueeze(sequence_mask(y_lengths, y.size(2)), 1).to(y.dtype)
    x = conv1d(sd, p + "pre", y) * x_mask
    x = wn_forward(sd, p + "enc.", x, x_mask, hidden, kernel_size, dilation_rate, num_layers, g=g)
    stats = conv1d(sd, p + "proj", x) * x_mask
    mean, log_scale = torch.split(stats, hidden, dim=1)
    if noise is None:
        noise = torch.randn_like(mean)
    z = (mean + noise * torch.exp(log_scale)) * x_mask
    return z, mean, log_scale, x_mask


def vits_voice_conversion(sd, y, y_lengths, g_src, g_tgt, args=None, noise=None):
    """Vits.voice_conversion, vits.py:1202-1228: posterior encoder -> flow (forward, source speaker) -> flow (reverse,
    target speaker) -> waveform decoder.  y [B, C_spec, T] linear spectrogram."""
    a = dict(VITS_DEFAULTS)
    a.update(args or {})
    h = a["hidden_channels"]
    z, _, _, y_mask = posterior_encoder(sd, "posterior_encoder.", y, y_lengths, h, a.get("kernel_size_posterior_encoder", 5),
                                        a.get("dilation_rate_posterior_encoder", 1),
                                        a.get("num_layers_posterior_encoder", 16), g=g_src, noise=noise)
    flow_cfg = dict(hidden=h, kernel_size=a["kernel_size_flow"], dilation_rate=a["dilation_rate_flow"],
                    num_layers=a["num_layers_flow"])
    z_p = residual_coupling_blocks_forward(sd, "flow.", z, y_mask, flow_cfg, g=g_src)
    z_hat = residual_coupling_blocks_reverse(sd, "flow.", z_p, y_mask, flow_cfg, g=g_tgt)
    o_hat = hifigan_forward(sd, "waveform_decoder.", z_hat * y_mask, vits_decoder_cfg(a), g=g_tgt)
    return {"model_outputs": o_hat, "y_mask": y_mask, "z": z, "z_p": z_p, "z_hat": z_hat}


def residual_coupling_blocks_reverse(sd, p, x, x_mask, cfg, g=None):
    """ResidualCouplingBlocks.forward(reverse=True), vits/networks.py:226-232."""
    for i in reversed(range(cfg.get("num_flows", 4))):
        x = torch.flip(x, [1])
        x = residual_coupling_reverse(sd, p + "flows.%d." % i, x, x_mask, cfg, g=g)
    return x


# ----------------------------------------------------------------------------------------------
# relative-position transformer — glow_tts/transformer.py
# ----------------------------------------------------------------------------------------------
def layer_norm2(sd, name, x, eps=1e-5):
    """LayerNorm2, generic/normalization.py:42-53 (torch layer_norm over channels)."""
    c = x.shape[1]
    return F.layer_norm(x.transpose(1, -1), (c,), sd[name + ".gamma"], sd[name + ".beta"], eps).transpose(1, -1)


def layer_norm1(sd, name, x, eps=1e-4):
    """LayerNorm, generic/normalization.py:23-28 (biased variance, rsqrt(var+eps), gamma/beta [1,C,1])."""
    mean = torch.mean(x, 1, keepdim=True)
    variance = torch.mean((x - mean) ** 2, 1, keepdim=True)
    x = (x - mean) * torch.rsqrt(variance + eps)
    return x * sd[name + ".gamma"].view(1, -1, 1) + sd[name + ".beta"].view(1, -1, 1)


def _get_relative_embeddings(emb, length, window):
    """transformer.py:196-207."""
    pad_length = max(length - (window + 1), 0)
    s = max((window + 1) - length, 0)
    e = s + 2 * length - 1
    if pad_length > 0:
        emb = F.pad(emb, [0, 0, pad_length, pad_length, 0, 0])
    return emb[:, s:e]


def _rel_to_abs(x):
    """transformer.py:209-225."""
    b, h, l, _ = x.size()
    x = F.pad(x, [0, 1, 0, 0, 0, 0, 0, 0])
    x_flat = x.view([b, h, l * 2 * l])
    x_flat = F.pad(x_flat, [0, l - 1, 0, 0, 0, 0])
    return x_flat.view([b, h, l + 1, 2 * l - 1])[:, :, :l, l - 1:]


def _abs_to_rel(x):
    """transformer.py:227-241."""
    b, h, l, _ = x.size()
    x = F.pad(x, [0, l - 1, 0, 0, 0, 0, 0, 0])
    x_flat = x.view([b, h, l ** 2 + l * (l - 1)])
    x_flat = F.pad(x_flat, [l, 0, 0, 0, 0, 0])
    return x_flat.view([b, h, l, 2 * l])[:, :, :, 1:]


def rel_mha(sd, p, x, attn_mask, num_heads, window):
    """RelativePositionMultiHeadAttention.forward/attention, transformer.py:106-163."""
    q = conv1d(sd, p + "conv_q", x)
    k = conv1d(sd, p + "conv_k", x)
    v = conv1d(sd, p + "conv_v", x)
    b, d, t = k.shape
    kc = d // num_heads
    q = q.view(b, num_heads, kc, t).transpose(2, 3)
    k = k.view(b, num_heads, kc, t).transpose(2, 3)
    v = v.view(b, num_heads, kc, t).transpose(2, 3)
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(kc)
    if window is not None:
        ek = _get_relative_embeddings(sd[p + "emb_rel_k"], t, window)
        rel_logits = torch.matmul(q, ek.unsqueeze(0).transpose(-2, -1))
        scores = scores + _rel_to_abs(rel_logits) / math.sqrt(kc)
    scores = scores.masked_fill(attn_mask == 0, -1e4)
    p_attn = F.softmax(scores, dim=-1)
    out = torch.matmul(p_attn, v)
    if window is not None:
        rw = _abs_to_rel(p_attn)
        ev = _get_relative_embeddings(sd[p + "emb_rel_v"], t, window)
        out = out + torch.matmul(rw, ev.unsqueeze(0))
    out = out.transpose(2, 3).contiguous().view(b, d, t)
    return conv1d(sd, p + "conv_o", out)


def ffn(sd, p, x, x_mask, kernel_size):
    """FeedForwardNetwork.forward, transformer.py:290-295 with _same_padding :306-313."""
    pl, pr = (kernel_size - 1) // 2, kernel_size // 2
    pad = (lambda t: t) if kernel_size == 1 else (lambda t: F.pad(t, [pl, pr]))
    x = conv1d(sd, p + "conv_1", pad(x * x_mask))
    x = torch.relu(x)
    x = conv1d(sd, p + "conv_2", pad(x * x_mask))
    return x * x_mask


def rel_pos_transformer(sd, p, x, x_mask, num_layers, num_heads, kernel_size, window, ln_type):
    """RelativePositionTransformer.forward, transformer.py:409-432."""
    ln = layer_norm2 if ln_type == "2" else layer_norm1
    attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    for i in range(num_layers):
        x = x * x_mask
        y = rel_mha(sd, p + "attn_layers.%d." % i, x, attn_mask, num_heads, window)
        x = ln(sd, p + "norm_layers_1.%d" % i, x + y)
        y = ffn(sd, p + "ffn_layers.%d." % i, x, x_mask, kernel_size)
        if (i + 1) == num_layers and (p + "proj.weight") in sd:
            x = conv1d(sd, p + "proj", x)
        x = ln(sd, p + "norm_layers_2.%d" % i, x + y)
    return x * x_mask


def text_encoder(sd, p, tokens, x_lengths, cfg, lang_emb=None):
    """TextEncoder.forward, vits/networks.py:79-100.  lang_emb [B,L,1] is concatenated as extra channels (:89-91)."""
    hidden = cfg["hidden_channels"]
    x = F.embedding(tokens, sd[p + "emb.weight"]) * math.sqrt(hidden)
    if lang_emb is not None:
        x = torch.cat((x, lang_emb.transpose(2, 1).expand(x.size(0), x.size(1), -1)), dim=-1)
    x = torch.transpose(x, 1, -1)
    x_mask = torch.unsqueeze(sequence_mask(x_lengths, x.size(2)), 1).to(x.dtype)
    x = rel_pos_transformer(sd, p + "encoder.", x * x_mask, x_mask, cfg["num_layers_text_encoder"],
                            cfg["num_heads_text_encoder"], cfg["kernel_size_text_encoder"], 4, "2")
    stats = conv1d(sd, p + "proj", x) * x_mask
    m, logs = torch.split(stats, hidden, dim=1)
    return x, m, logs, x_mask


# ----------------------------------------------------------------------------------------------
# stochastic duration predictor — vits/stochastic_duration_predictor.py, vits/transforms.py
# ----------------------------------------------------------------------------------------------
def dds_conv(sd, p, x, x_mask, kernel_size, num_layers, g=None):
    """DilatedDepthSeparableConv.forward, stochastic_duration_predictor.py:46-63."""
    if g is not None:
        x = x + g
    c = x.shape[1]
    for i in range(num_layers):
        d = kernel_size ** i
        y = conv1d(sd, p + "convs_sep.%d" % i, x * x_mask, dilation=d, padding=(kernel_size * d - d) // 2, groups=c)
        y = layer_norm2(sd, p + "norms_1.%d" % i, y)
        y = F.gelu(y)
        y = conv1d(sd, p + "convs_1x1.%d" % i, y)
        y = layer_norm2(sd, p + "norms_2.%d" % i, y)
        y = F.gelu(y)
        x = x + y
    return x * x_mask


def rq_spline_inverse(inputs, uw, uh, ud, tail_bound=5.0, min_bin_width=1e-3, min_bin_height=1e-3,
                      min_derivative=1e-3):
    """unconstrained_rational_quadratic_spline(inverse=True, tails='linear') + rational_quadratic_spline,
    vits/transforms.py:50-184.  Vectorised with `where` instead of boolean-mask scatter; elementwise
    arithmetic and its order follow the reference exactly."""
    inside = (inputs >= -tail_bound) & (inputs <= tail_bound)           # transforms.py:62
    ud = F.pad(ud, pad=(1, 1))
    constant = np.log(np.exp(1 - min_derivative) - 1)                     # transforms.py:70
    ud[..., 0] = constant
    ud[..., -1] = constant
    num_bins = uw.shape[-1]
    left = bottom = -tail_bound
    right = top = tail_bound
    x_in = torch.where(inside, inputs, torch.zeros_like(inputs))         # dummy in-domain value outside

    widths = F.softmax(uw, dim=-1)                                        # transforms.py:122-129
    widths = min_bin_width + (1 - min_bin_width * num_bins) * widths
    cumwidths = torch.cumsum(widths, dim=-1)
    cumwidths = F.pad(cumwidths, pad=(1, 0), mode="constant", value=0.0)
    cumwidths = (right - left) * cumwidths + left
    cumwidths[..., 0] = left
    cumwidths[..., -1] = right
    widths = cumwidths[..., 1:] - cumwidths[..., :-1]
    derivatives = min_derivative + F.softplus(ud)                         # transforms.py:131
    heights = F.softmax(uh, dim=-1)                                       # transforms.py:133-140
    heights = min_bin_height + (1 - min_bin_height * num_bins) * heights
    cumheights = torch.cumsum(heights, dim=-1)
    cumheights = F.pad(cumheights, pad=(1, 0), mode="constant", value=0.0)
    cumheights = (top - bottom) * cumheights + bottom
    cumheights[..., 0] = bottom
    cumheights[..., -1] = top
    heights = cumheights[..., 1:] - cumheights[..., :-1]

    locs = cumheights.clone()                                             # searchsorted, transforms.py:45-47
    locs[..., -1] += 1e-6
    bin_idx = (torch.sum(x_in[..., None] >= locs, dim=-1) - 1)[..., None]

    input_cumwidths = cumwidths.gather(-1, bin_idx)[..., 0]
    input_bin_widths = widths.gather(-1, bin_idx)[..., 0]
    input_cumheights = cumheights.gather(-1, bin_idx)[..., 0]
    delta = heights / widths
    input_delta = delta.gather(-1, bin_idx)[..., 0]
    input_derivatives = derivatives.gather(-1, bin_idx)[..., 0]
    input_derivatives_plus_one = derivatives[..., 1:].gather(-1, bin_idx)[..., 0]
    input_heights = heights.gather(-1, bin_idx)[..., 0]

    a = (x_in - input_cumheights) * (input_derivatives + input_derivatives_plus_one - 2 * input_delta) \
        + input_heights * (input_delta - input_derivatives)             # transforms.py:159-165
    b = input_heights * input_derivatives - (x_in - input_cumheights) * (
        input_derivatives + input_derivatives_plus_one - 2 * input_delta)
    c = -input_delta * (x_in - input_cumheights)
    discriminant = b.pow(2) - 4 * a * c
    root = (2 * c) / (-b - torch.sqrt(discriminant))
    outputs = root * input_bin_widths + input_cumwidths
    return torch.where(inside, outputs, inputs)


def conv_flow_reverse(sd, p, x, x_mask, g, hidden, kernel_size, num_layers, num_bins=10, tail_bound=5.0):
    """ConvFlow.forward(reverse=True), stochastic_duration_predictor.py:120-147 (in_channels=2)."""
    x0, x1 = x[:, :1], x[:, 1:]
    h = conv1d(sd, p + "pre", x0)
    h = dds_conv(sd, p + "convs.", h, x_mask, kernel_size, num_layers, g=g)
    h = conv1d(sd, p + "proj", h) * x_mask
    b, c, t = x0.shape
    h = h.reshape(b, c, -1, t).permute(0, 1, 3, 2)
    uw = h[..., :num_bins] / math.sqrt(hidden)
    uh = h[..., num_bins:2 * num_bins] / math.sqrt(hidden)
    ud = h[..., 2 * num_bins:]
    x1 = rq_spline_inverse(x1, uw, uh, ud, tail_bound=tail_bound)
    return torch.cat([x0, x1], 1) * x_mask


def sdp_reverse(sd, p, x, x_mask, noise, noise_scale=1.0, hidden=192, kernel_size=3, num_flows=4, g=None, lang_emb=None):
    """StochasticDurationPredictor.forward(reverse=True), stochastic_duration_predictor.py:230-239,283-294.

    `noise` [B,2,T] replaces the internal torch.randn (:287) so both sides see identical draws."""
    x = conv1d(sd, p + "pre", x)
    if g is not None:
        x = x + conv1d(sd, p + "cond", g)
    if lang_emb is not None:                                   # :235-236
        x = x + conv1d(sd, p + "cond_lang", lang_emb)
    x = dds_conv(sd, p + "convs.", x, x_mask, kernel_size, 3)
    x = conv1d(sd, p + "proj", x) * x_mask
    order = list(reversed(range(num_flows + 1)))      # flows[4],[3],[2],[1],[0]
    order = order[:-2] + [order[-1]]                  # "remove a useless vflow" (:285-286)
    z = noise * noise_scale
    for i in order:
        z = torch.flip(z, [1])
        if i == 0:  # ElementwiseAffine reverse (:82-83)
            z = (z - sd[p + "flows.0.translation"]) * torch.exp(-sd[p + "flows.0.log_scale"]) * x_mask
        else:
            z = conv_flow_reverse(sd, p + "flows.%d." % i, z, x_mask, x, hidden, kernel_size, 3)
    return z[:, :1]


def duration_predictor(sd, p, x, x_mask, g=None, lang_emb=None):
    """glow_tts/duration_predictor.py:46-69 (conv -> relu -> LayerNorm(1e-4), twice, then 1x1)."""
    if g is not None:
        x = x + conv1d(sd, p + "cond", g)
    if lang_emb is not None:                                   # :61-62
        x = x + conv1d(sd, p + "cond_lang", lang_emb)
    k = weight(sd, p + "conv_1").shape[-1]
    x = conv1d(sd, p + "conv_1", x * x_mask, padding=k // 2)
    x = layer_norm1(sd, p + "norm_1", torch.relu(x))
    x = conv1d(sd, p + "conv_2", x * x_mask, padding=k // 2)
    x = layer_norm1(sd, p + "norm_2", torch.relu(x))
    x = conv1d(sd, p + "proj", x * x_mask)
    return x * x_mask


# ----------------------------------------------------------------------------------------------
# VITS inference glue — TTS/tts/models/vits.py:1088-1173
# ----------------------------------------------------------------------------------------------
VITS_DEFAULTS = dict(  # VitsArgs, vits.py:544-600
    num_chars=100, hidden_channels=192, hidden_channels_ffn_text_encoder=768, num_heads_text_encoder=2,
    num_layers_text_encoder=6, kernel_size_text_encoder=3, kernel_size_flow=5, dilation_rate_flow=1,
    num_layers_flow=4, resblock_type_decoder="1", resblock_kernel_sizes_decoder=[3, 7, 11],
    resblock_dilation_sizes_decoder=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], upsample_rates_decoder=[8, 8, 2, 2],
    upsample_initial_channel_decoder=512, upsample_kernel_sizes_decoder=[16, 16, 4, 4], use_sdp=True,
    inference_noise_scale=0.667, length_scale=1.0, inference_noise_scale_dp=1.0, max_inference_len=None,
)


def vits_decoder_cfg(args):
    return dict(resblock_type=args["resblock_type_decoder"],
                resblock_dilation_sizes=args["resblock_dilation_sizes_decoder"],
                resblock_kernel_sizes=args["resblock_kernel_sizes_decoder"],
                upsample_kernel_sizes=args["upsample_kernel_sizes_decoder"],
                upsample_factors=args["upsample_rates_decoder"], inference_padding=0)


def vits_speaker_g(sd, speaker_ids=None, d_vectors=None):
    """vits.py:873-886,1116-1117: g [B,C,1] from the speaker-embedding table or L2-normalised d-vectors."""
    if d_vectors is not None:
        return F.normalize(d_vectors).unsqueeze(-1)
    if speaker_ids is not None:
        return F.embedding(speaker_ids, sd["emb_g.weight"]).unsqueeze(-1)
    return None


def vits_language_emb(sd, language_ids):
    """vits.py:1119-1122: lang_emb = emb_l(lid).unsqueeze(-1)."""
    return None if language_ids is None else F.embedding(language_ids, sd["emb_l.weight"]).unsqueeze(-1)


def interpolate_vocoder_input(scale_factor, spec):
    """TTS/vocoder/utils/generic_utils.py:11-29 (called at synthesizer.py:418-424 when the TTS and vocoder sample rates
    differ): spec numpy [C, T] -> tensor [1, C, T'] by bilinear interpolation with scale_factor = [1, sr_vocoder / sr_tts],
    recompute_scale_factor=True, align_corners=False."""
    spec = torch.tensor(spec).unsqueeze(0).unsqueeze(0)
    return F.interpolate(spec, scale_factor=scale_factor, recompute_scale_factor=True, mode="bilinear",
                         align_corners=False).squeeze(0)


def vits_inference(sd, tokens, x_lengths=None, args=None, noise_dp=None, noise_z=None, durations=None,
                   stop_after=None, g=None, lang_emb=None):
    """Vits.inference, vits.py:1088-1173 (g: speaker conditioning [B,C,1]; lang_emb: language embedding [B,L,1]).

    noise_dp [B,2,T_text] / noise_z [B,C,T_dec] replace the internal randn draws
    (stochastic_duration_predictor.py:287, vits.py:1155); when None they are drawn with torch.randn in
    the reference's order.  `durations` [B,1,T_text] overrides w_ceil (parity runs inject the other
    side's integer durations — SURVEY §7 "ceil() cliff")."""
    a = dict(VITS_DEFAULTS)
    a.update(args or {})
    if x_lengths is None:
        x_lengths = torch.tensor(tokens.shape[1:2])                         # vits.py:1082-1086
    x, m_p, logs_p, x_mask = text_encoder(sd, "text_encoder.", tokens, x_lengths, a, lang_emb=lang_emb)
    out = {"x": x, "m_p_text": m_p, "logs_p_text": logs_p, "x_mask": x_mask}
    if durations is None:
        if a["use_sdp"]:
            if noise_dp is None:
                noise_dp = torch.randn(x.size(0), 2, x.size(2))
            logw = sdp_reverse(sd, "duration_predictor.", x, x_mask, noise_dp, a["inference_noise_scale_dp"],
                               hidden=192, g=g if a.get("condition_dp_on_speaker", True) else None, lang_emb=lang_emb)
        else:
            logw = duration_predictor(sd, "duration_predictor.", x, x_mask,
                                      g=g if a.get("condition_dp_on_speaker", True) else None, lang_emb=lang_emb)
        out["logw"] = logw
        w = torch.exp(logw) * x_mask * a["length_scale"]                    # vits.py:1140
        w_ceil = torch.ceil(w)
    else:
        w_ceil = durations
    y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()        # vits.py:1146
    y_mask = sequence_mask(y_lengths, None).to(x_mask.dtype).unsqueeze(1)
    attn_mask = x_mask * y_mask.transpose(1, 2)
    attn = generate_path(w_ceil.squeeze(1), attn_mask.squeeze(1).transpose(1, 2))
    m_p = torch.matmul(attn.transpose(1, 2), m_p.transpose(1, 2)).transpose(1, 2)
    logs_p = torch.matmul(attn.transpose(1, 2), logs_p.transpose(1, 2)).transpose(1, 2)
    if noise_z is None:
        noise_z = torch.randn_like(m_p)
    z_p = m_p + noise_z * torch.exp(logs_p) * a["inference_noise_scale"]    # vits.py:1155
    out.update(durations=w_ceil, y_mask=y_mask, alignments=attn, m_p=m_p, logs_p=logs_p, z_p=z_p,
               y_lengths=y_lengths)
    if stop_after == "prior":
        return out
    flow_cfg = dict(hidden=a["hidden_channels"], kernel_size=a["kernel_size_flow"],
                    dilation_rate=a["dilation_rate_flow"], num_layers=a["num_layers_flow"])
    z = residual_coupling_blocks_reverse(sd, "flow.", z_p, y_mask, flow_cfg, g=g)
    out["z"] = z
    if stop_after == "flow":
        return out
    o = hifigan_forward(sd, "waveform_decoder.", (z * y_mask)[:, :, : a["max_inference_len"]], vits_decoder_cfg(a), g=g)
    out["model_outputs"] = o
    return out


# ----------------------------------------------------------------------------------------------
# Glow-TTS — TTS/tts/layers/glow_tts/{encoder,decoder,glow}.py, TTS/tts/models/glow_tts.py
# ----------------------------------------------------------------------------------------------
GLOW_DEFAULTS = dict(  # GlowTTSConfig, glow_tts_config.py:101-152
    num_chars=130, hidden_channels_enc=192, hidden_channels_dec=192, hidden_channels_dp=256, out_channels=80,
    num_flow_blocks_dec=12, kernel_size_dec=5, dilation_rate=1, num_block_layers=4, num_splits=4, num_squeeze=2,
    sigmoid_scale=False, mean_only=True, use_encoder_prenet=True, inference_noise_scale=0.0, length_scale=1.0,
    encoder_params=dict(kernel_size=3, num_layers=6, num_heads=2, hidden_channels_ffn=768,
                        rel_attn_window_size=None, layer_norm_type="1"),
)


def glow_prenet(sd, p, x, x_mask, num_layers=3, kernel_size=5):
    """ResidualConv1dLayerNormBlock.forward, glow_tts/glow.py:55-67."""
    x_res = x
    for i in range(num_layers):
        x = conv1d(sd, p + "conv_layers.%d" % i, x * x_mask, padding=kernel_size // 2)
        x = layer_norm1(sd, p + "norm_layers.%d" % i, x * x_mask)
        x = F.relu(x)
    x = x_res + conv1d(sd, p + "proj", x)
    return x * x_mask


def glow_speaker_g(sd, speaker_ids=None, d_vectors=None):
    """GlowTTS._speaker_embedding, glow_tts.py:179-191: L2-normalised table row or d-vector, [B,C,1]."""
    if speaker_ids is not None:
        return F.normalize(F.embedding(speaker_ids, sd["emb_g.weight"])).unsqueeze(-1)
    if d_vectors is not None:
        return F.normalize(d_vectors).unsqueeze(-1)
    return None


def glow_encoder(sd, p, tokens, x_lengths, a, g=None):
    """Encoder.forward (rel_pos_transformer type), glow_tts/encoder.py:143-179.  g [B,C,1]: concatenated to the
    duration predictor's input over time (:166-168)."""
    hidden = a["hidden_channels_enc"]
    ep = a["encoder_params"]
    x = F.embedding(tokens, sd[p + "emb.weight"]) * math.sqrt(hidden)
    x = torch.transpose(x, 1, -1)
    x_mask = torch.unsqueeze(sequence_mask(x_lengths, x.size(2)), 1).to(x.dtype)
    if a["use_encoder_prenet"]:
        x = glow_prenet(sd, p + "prenet.", x, x_mask)
    x = rel_pos_transformer(sd, p + "encoder.", x, x_mask, ep["num_layers"], ep["num_heads"], ep["kernel_size"],
                            ep.get("rel_attn_window_size"), ep.get("layer_norm_type", "1"))
    x_m = conv1d(sd, p + "proj_m", x) * x_mask
    if not a["mean_only"]:
        x_logs = conv1d(sd, p + "proj_s", x) * x_mask
    else:
        x_logs = torch.zeros_like(x_m)
    x_dp = x if g is None else torch.cat([x, g.expand(-1, -1, x.size(-1))], 1)
    logw = duration_predictor(sd, p + "duration_predictor.", x_dp, x_mask)
    return x_m, x_logs, logw, x_mask


def glow_squeeze(x, x_mask, n=2):
    """glow_tts/decoder.py:8-28."""
    b, c, t = x.size()
    t = (t // n) * n
    x = x[:, :, :t]
    x_sqz = x.view(b, c, t // n, n).permute(0, 3, 1, 2).contiguous().view(b, c * n, t // n)
    x_mask = x_mask[:, :, n - 1::n]
    return x_sqz * x_mask, x_mask


def glow_unsqueeze(x, x_mask, n=2):
    """glow_tts/decoder.py:31-47."""
    b, c, t = x.size()
    x_unsqz = x.view(b, n, c // n, t).permute(0, 2, 3, 1).contiguous().view(b, c // n, t * n)
    x_mask = x_mask.unsqueeze(-1).repeat(1, 1, 1, n).view(b, 1, t * n)
    return x_unsqz * x_mask, x_mask


def glow_decoder_reverse(sd, p, x, x_mask, a, g=None):
    """Decoder.forward(reverse=True), glow_tts/decoder.py:113-137 with ActNorm (normalization.py:98-101),
    InvConvNear (glow.py:107-137) and CouplingBlock (glow.py:200-229) in reverse."""
    ns, nsq = a["num_splits"], a["num_squeeze"]
    if nsq > 1:
        x, x_mask = glow_squeeze(x, x_mask, nsq)
    cin = x.shape[1]
    for blk in reversed(range(a["num_flow_blocks_dec"])):
        pa, pi, pc = (p + "flows.%d." % (3 * blk + j) for j in range(3))
        # CouplingBlock reverse
        x0, x1 = x[:, : cin // 2], x[:, cin // 2:]
        h = conv1d(sd, pc + "start", x0) * x_mask
        h = wn_forward(sd, pc + "wn.", h, x_mask, a["hidden_channels_dec"], a["kernel_size_dec"],
                       a["dilation_rate"], a["num_block_layers"], g=g)
        out = conv1d(sd, pc + "end", h)
        t_, s_ = out[:, : cin // 2], out[:, cin // 2:]
        if a["sigmoid_scale"]:
            s_ = torch.log(1e-6 + torch.sigmoid(s_ + 2))
        x = torch.cat([x0, (x1 - t_) * torch.exp(-s_) * x_mask], 1)
        # InvConvNear reverse
        b, c, t = x.size()
        w = sd[pi + "weight_inv"] if (pi + "weight_inv") in sd else torch.inverse(sd[pi + "weight"].float())
        xx = x.view(b, 2, c // ns, ns // 2, t).permute(0, 1, 3, 2, 4).contiguous().view(b, ns, c // ns, t)
        z = F.conv2d(xx, w.view(ns, ns, 1, 1))
        x = z.view(b, 2, ns // 2, c // ns, t).permute(0, 1, 3, 2, 4).contiguous().view(b, c, t) * x_mask
        # ActNorm reverse
        x = (x - sd[pa + "bias"]) * torch.exp(-sd[pa + "logs"]) * x_mask
    if nsq > 1:
        x, x_mask = glow_unsqueeze(x, x_mask, nsq)
    return x


def glow_decoder_forward(sd, p, x, x_mask, a, g=None):
    """Decoder.forward(reverse=False), glow_tts/decoder.py:113-137: per block ActNorm (normalization.py:102-103),
    InvConvNear with the weight itself (glow.py:107-137), CouplingBlock forward (glow.py:200-226)."""
    ns, nsq = a["num_splits"], a["num_squeeze"]
    if nsq > 1:
        x, x_mask = glow_squeeze(x, x_mask, nsq)
    cin = x.shape[1]
    for blk in range(a["num_flow_blocks_dec"]):
        pa, pi, pc = (p + "flows.%d." % (3 * blk + j) for j in range(3))
        x = (sd[pa + "bias"] + torch.exp(sd[pa + "logs"]) * x) * x_mask
        b, c, t = x.size()
        xx = x.view(b, 2, c // ns, ns // 2, t).permute(0, 1, 3, 2, 4).contiguous().view(b, ns, c // ns, t)
        z = F.conv2d(xx, sd[pi + "weight"].view(ns, ns, 1, 1))
        x = z.view(b, 2, ns // 2, c // ns, t).permute(0, 1, 3, 2, 4).contiguous().view(b, c, t) * x_mask
        x0, x1 = x[:, : cin // 2], x[:, cin // 2:]
        h = conv1d(sd, pc + "start", x0) * x_mask
        h = wn_forward(sd, pc + "wn.", h, x_mask, a["hidden_channels_dec"], a["kernel_size_dec"], a["dilation_rate"],
                       a["num_block_layers"], g=g)
        out = conv1d(sd, pc + "end", h)
        t_, s_ = out[:, : cin // 2], out[:, cin // 2:]
        x = torch.cat([x0, (t_ + torch.exp(s_) * x1) * x_mask], 1)
    if nsq > 1:
        x, x_mask = glow_unsqueeze(x, x_mask, nsq)
    return x


def glow_inference_with_mas(sd, tokens, x_lengths, y, y_lengths, args=None, maximum_path=None, g=None):
    """GlowTTS.inference_with_MAS, glow_tts.py:262-316.  y [B,T,C] mel.  `maximum_path(value, mask)` = the MAS
    implementation to use (the test passes the C oracle)."""
    a = dict(GLOW_DEFAULTS)
    a.update(args or {})
    y = y.transpose(1, 2)
    o_mean, o_log_scale, o_dur_log, x_mask = glow_encoder(sd, "encoder.", tokens, x_lengths, a, g=g)
    n = a["num_squeeze"]
    y = y[:, :, : (y.size(2) // n) * n]
    y_lengths = torch.div(y_lengths, n, rounding_mode="floor") * n
    y_mask = torch.unsqueeze(sequence_mask(y_lengths, y.size(2)), 1).to(x_mask.dtype)
    attn_mask = torch.unsqueeze(x_mask, -1) * torch.unsqueeze(y_mask, 2)
    z = glow_decoder_forward(sd, "decoder.", y, y_mask, a, g=g)
    logp = mas_logp(z, o_mean, o_log_scale, glow_order=True)
    attn = maximum_path(logp, attn_mask.squeeze(1))
    y_mean = torch.matmul(attn.transpose(1, 2), o_mean.transpose(1, 2)).transpose(1, 2)
    o_attn_dur = torch.log(1 + torch.sum(attn, -1)).unsqueeze(1) * x_mask
    zz = y_mean * y_mask
    return {"model_outputs": zz.transpose(1, 2), "z": z, "logp": logp, "alignments": attn.permute(0, 2, 1),
            "y_mean": y_mean.transpose(1, 2), "total_durations_log": o_attn_dur.transpose(1, 2)}


def glow_tts_inference(sd, tokens, x_lengths, args=None, noise=None, g=None, durations=None):
    """GlowTTS.inference, glow_tts.py:341-374 (g: speaker conditioning from glow_speaker_g, or None).
    `durations` [B,1,T] (not a reference argument) replaces w_ceil: benches / parity harnesses pin the output length."""
    a = dict(GLOW_DEFAULTS)
    a.update(args or {})
    o_mean, o_log_scale, o_dur_log, x_mask = glow_encoder(sd, "encoder.", tokens, x_lengths, a, g=g)
    w = (torch.exp(o_dur_log) - 1) * x_mask * a["length_scale"]
    w_ceil = torch.clamp_min(torch.ceil(w), 1)
    if durations is not None:
        w_ceil = durations.to(w.dtype) * x_mask
    y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
    y_mask = torch.unsqueeze(sequence_mask(y_lengths, None), 1).to(x_mask.dtype)
    attn_mask = torch.unsqueeze(x_mask, -1) * torch.unsqueeze(y_mask, 2)
    attn = generate_path(w_ceil.squeeze(1), attn_mask.squeeze(1)).unsqueeze(1)
    y_mean = torch.matmul(attn.squeeze(1).transpose(1, 2), o_mean.transpose(1, 2)).transpose(1, 2)
    y_log_scale = torch.matmul(attn.squeeze(1).transpose(1, 2), o_log_scale.transpose(1, 2)).transpose(1, 2)
    if noise is None:
        noise = torch.randn_like(y_mean)
    z = (y_mean + torch.exp(y_log_scale) * noise * a["inference_noise_scale"]) * y_mask
    y = glow_decoder_reverse(sd, "decoder.", z, y_mask, a, g=g)
    return {"model_outputs": y.transpose(1, 2), "alignments": attn.squeeze(1).permute(0, 2, 1),
            "durations_log": o_dur_log.transpose(1, 2), "y_mean": y_mean.transpose(1, 2), "durations": w_ceil,
            "y_lengths": y_lengths}


def mas_logp(z, m, logs, glow_order=False):
    """vits.py:912-918 (glow_tts.py:241-247 when glow_order): log-likelihood matrix [B,T_x,T_y] fed to MAS."""
    o_scale = torch.exp(-2 * logs)
    logp1 = torch.sum(-0.5 * math.log(2 * math.pi) - logs, [1]).unsqueeze(-1)
    logp2 = torch.einsum("klm, kln -> kmn", [o_scale, -0.5 * (z ** 2)])
    logp3 = torch.einsum("klm, kln -> kmn", [m * o_scale, z])
    logp4 = torch.sum(-0.5 * (m ** 2) * o_scale, [1]).unsqueeze(-1)
    return (logp1 + logp2 + logp3 + logp4) if glow_order else (logp2 + logp3 + logp1 + logp4)


def vits_forward_mas(z_p, m_p, logs_p, x_mask, y_mask, maximum_path, use_sdp=True):
    """Vits.forward_mas, vits.py:909-936 — the alignment half (the duration LOSS of :921-941 is training code):
    attn_mask, logp, `attn = maximum_path(logp, attn_mask.squeeze(1)).unsqueeze(1)`, `attn_durations = attn.sum(3)` and,
    for the deterministic predictor, `attn_log_durations = log(attn_durations + 1e-6) * x_mask`.
    x_mask [B,1,T_x], y_mask [B,1,T_y]; maximum_path(value [B,T_x,T_y], mask) -> path (test infrastructure: the C oracle)."""
    attn_mask = torch.unsqueeze(x_mask, -1) * torch.unsqueeze(y_mask, 2)
    logp = mas_logp(z_p, m_p, logs_p, glow_order=False)
    attn = maximum_path(logp, attn_mask.squeeze(1)).unsqueeze(1)
    out = {"logp": logp, "attn": attn, "attn_durations": attn.sum(3)}
    if not use_sdp:
        out["attn_log_durations"] = torch.log(out["attn_durations"] + 1e-6) * x_mask
    return out


def vits_forward_align(sd, tokens, x_lengths, y, y_lengths, maximum_path, args=None, noise=None, g=None, lang_emb=None):
    """The alignment pass of Vits.forward, vits.py:1018-1031: text encoder -> posterior encoder -> flow (forward) ->
    forward_mas -> prior expanded along the path (`einsum("klmn, kjm -> kjn", attn, m_p)`).  y [B, C_spec, T_y]."""
    a = dict(VITS_DEFAULTS)
    a.update(args or {})
    h = a["hidden_channels"]
    x, m_p, logs_p, x_mask = text_encoder(sd, "text_encoder.", tokens, x_lengths, a, lang_emb=lang_emb)
    z, m_q, logs_q, y_mask = posterior_encoder(sd, "posterior_encoder.", y, y_lengths, h, a.get("kernel_size_posterior_encoder", 5),
                                               a.get("dilation_rate_posterior_encoder", 1),
                                               a.get("num_layers_posterior_encoder", 16), g=g, noise=noise)
    flow_cfg = dict(hidden=h, kernel_size=a["kernel_size_flow"], dilation_rate=a["dilation_rate_flow"],
                    num_layers=a["num_layers_flow"])
    z_p = residual_coupling_blocks_forward(sd, "flow.", z, y_mask, flow_cfg, g=g)
    out = vits_forward_mas(z_p, m_p, logs_p, x_mask, y_mask, maximum_path, use_sdp=a.get("use_sdp", True))
    out.update(x=x, z=z, z_p=z_p, m_q=m_q, logs_q=logs_q, x_mask=x_mask, y_mask=y_mask,
               m_p=torch.einsum("klmn, kjm -> kjn", [out["attn"], m_p]),
               logs_p=torch.einsum("klmn, kjm -> kjn", [out["attn"], logs_p]))
    return out


# ----------------------------------------------------------------------------------------------
# seeded weight factory: lives in the product package (bench.py needs synthetic checkpoints too and may not
# import oracle/); re-exported here for the tests.
# ----------------------------------------------------------------------------------------------
from tts_amd.synthetic import make_hifigan_state  # noqa: E402,F401
