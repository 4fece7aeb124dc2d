"""Host-side layer drivers of the VITS / Glow-TTS acoustic models over the HIP kernels.

Each class mirrors one reference `nn.Module` (same parameter names read from a reference-layout
`state_dict`, same forward semantics) but owns no arithmetic: `__init__` folds/re-orders/packs the
weights once (load-time glue), `__call__` is a fixed sequence of kernel launches through the C ABI
(tts_amd.ops).  torch tensors are device buffers only.
"""
import math

import torch

from . import ops
from .ops import (ACT_GELU, ACT_NONE, ACT_RELU, CONV_COUPLE, CONV_COUPLE_AFFINE, CONV_COUPLE_AFFINE_FWD, CONV_COUPLE_AFFINE_MIX, CONV_GATE, CONV_RES_SKIP,
                  PackedConv, fold_weight_norm)


def _dev(t, device):
    return t.detach().to(device, torch.float32).contiguous()


def _new(like, c=None, t=None):
    B, C, T = like.shape
    return torch.empty((B, C if c is None else c, T if t is None else t), dtype=torch.float32, device=like.device)


class _Norm:
    """gamma/beta of LayerNorm ([1,C,1], eps 1e-4, normalization.py:5-28) or LayerNorm2 ([C], eps 1e-5, :31-53)."""

    def __init__(self, sd, name, device, eps):
        self.gamma = _dev(sd[name + ".gamma"].reshape(-1), device)
        self.beta = _dev(sd[name + ".beta"].reshape(-1), device)
        self.eps = eps


# ------------------------------------------------------------------------------------------------
# relative-position transformer — TTS/tts/layers/glow_tts/transformer.py:322-432
# ------------------------------------------------------------------------------------------------
class RelativePositionTransformer:
    def __init__(self, sd, p, device, num_layers, num_heads, kernel_size, rel_attn_window_size, layer_norm_type):
        self.num_layers, self.num_heads, self.window = num_layers, num_heads, rel_attn_window_size
        eps = 1e-5 if str(layer_norm_type) == "2" else 1e-4
        self.layers = []
        for i in range(num_layers):
            a = p + "attn_layers.%d." % i
            L = {}
            wq, wk, wv = (fold_weight_norm(sd, a + n) for n in ("conv_q", "conv_k", "conv_v"))
            bq, bk, bv = (sd[a + n + ".bias"] for n in ("conv_q", "conv_k", "conv_v"))
            # one fused projection launch: rows q | k | v  (transformer.py:106-110)
            L["qkv"] = PackedConv(torch.cat([wq, wk, wv], 0), torch.cat([bq, bk, bv], 0), device)
            L["o"] = PackedConv(fold_weight_norm(sd, a + "conv_o"), sd[a + "conv_o.bias"], device)
            if rel_attn_window_size is not None:
                L["emb_k"] = _dev(sd[a + "emb_rel_k"][0], device)   # heads_share=True: [1, 2w+1, dk]
                L["emb_v"] = _dev(sd[a + "emb_rel_v"][0], device)
            L["n1"] = _Norm(sd, p + "norm_layers_1.%d" % i, device, eps)
            f = p + "ffn_layers.%d." % i
            L["f1"] = PackedConv(fold_weight_norm(sd, f + "conv_1"), sd[f + "conv_1.bias"], device,
                                 pad_left=(kernel_size - 1) // 2)
            L["f2"] = PackedConv(fold_weight_norm(sd, f + "conv_2"), sd[f + "conv_2.bias"], device,
                                 pad_left=(kernel_size - 1) // 2)
            L["n2"] = _Norm(sd, p + "norm_layers_2.%d" % i, device, eps)
            self.layers.append(L)
        self.proj = None
        if (p + "proj.weight") in sd:  # only when out_channels != hidden (transformer.py:398-399,426-427)
            self.proj = PackedConv(sd[p + "proj.weight"], sd.get(p + "proj.bias"), device)

    def __call__(self, x, mask):
        """x [B,H,T] already multiplied by mask, mask [B,T] -> [B,H_out,T] (masked).  transformer.py:409-432.
        Padded columns are zeroed after every norm (the reference zeroes them at the next layer's entry
        and at the end; valid columns are identical)."""
        B, H, T = x.shape
        for i, L in enumerate(self.layers):
            qkv = _new(x, 3 * H)
            ops.conv1d(L["qkv"], x, qkv)
            att = _new(x)
            ops.rel_attention(qkv, att, mask, self.num_heads, L.get("emb_k"), L.get("emb_v"), self.window or 0)
            xy = _new(x)
            ops.conv1d(L["o"], att, xy, res=x)                                  # x + attn(x)
            x1 = ops.channel_norm(xy, _new(x), L["n1"].gamma, L["n1"].beta, L["n1"].eps, out_mask=mask)
            hid = _new(x, L["f1"].c_out)
            # (t_out = T: an even FFN kernel pads (k-1)//2 left, k//2 right — transformer.py:306-313 — and keeps the length)
            ops.conv1d(L["f1"], x1, hid, out_act=ACT_RELU, out_mask=mask, t_out=T)       # relu(conv_1(x*mask)) * mask
            last = (i + 1) == self.num_layers
            if last and self.proj is not None:
                xo = _new(x, self.proj.c_out)
                ops.conv1d(self.proj, x1, xo)
                y = _new(xo)
                ops.conv1d(L["f2"], hid, y, out_mask=mask, t_out=T)
                x = ops.channel_norm(y, _new(xo), L["n2"].gamma, L["n2"].beta, L["n2"].eps, pre_res=xo, out_mask=mask)
            else:
                y = _new(x)
                ops.conv1d(L["f2"], hid, y, res=x1, out_mask=mask, t_out=T)    # x + ffn(x) (x1 is masked)
                x = ops.channel_norm(y, _new(x), L["n2"].gamma, L["n2"].beta, L["n2"].eps, out_mask=mask)
        return x


# ------------------------------------------------------------------------------------------------
# VITS text encoder — TTS/tts/layers/vits/networks.py:29-100
# ------------------------------------------------------------------------------------------------
class TextEncoder:
    def __init__(self, sd, p, device, hidden, num_layers, num_heads, kernel_size):
        self.hidden = hidden
        self.emb = _dev(sd[p + "emb.weight"], device)
        self.encoder = RelativePositionTransformer(sd, p + "encoder.", device, num_layers, num_heads, kernel_size, 4, "2")
        self.proj = PackedConv(sd[p + "proj.weight"], sd[p + "proj.bias"], device)

    def __call__(self, tokens, x_mask, lang=None):
        """tokens int64 [B,T], x_mask [B,T] -> x [B,H,T], stats [B,2H,T] (m | logs), networks.py:79-100.
        lang [B,L] (language embedding, multilingual models): appended as L extra channels, so the encoder runs at
        H+L channels and x is [B,H+L,T] (networks.py:62-63,89-91)."""
        B, T = tokens.shape
        if lang is None:
            x = torch.empty((B, self.hidden, T), dtype=torch.float32, device=tokens.device)
            ops.embed(tokens, self.emb, x_mask, math.sqrt(self.hidden), x)
        else:
            x = torch.empty((B, self.hidden + lang.shape[1], T), dtype=torch.float32, device=tokens.device)
            ops.embed_cat(tokens, self.emb, x_mask, math.sqrt(self.hidden), lang, x)
        x = self.encoder(x, x_mask)
        stats = _new(x, self.proj.c_out)
        ops.conv1d(self.proj, x, stats, out_mask=x_mask)
        return x, stats


# ------------------------------------------------------------------------------------------------
# DilatedDepthSeparableConv — TTS/tts/layers/vits/stochastic_duration_predictor.py:11-63
# ------------------------------------------------------------------------------------------------
class DDSConv:
    def __init__(self, sd, p, device, channels, kernel_size, num_layers):
        self.layers = []
        for i in range(num_layers):
            self.layers.append(dict(
                dw_w=_dev(sd[p + "convs_sep.%d.weight" % i].reshape(channels, kernel_size), device),
                dw_b=_dev(sd[p + "convs_sep.%d.bias" % i], device), dil=kernel_size ** i,
                pw=PackedConv(sd[p + "convs_1x1.%d.weight" % i], sd[p + "convs_1x1.%d.bias" % i], device),
                n1=_Norm(sd, p + "norms_1.%d" % i, device, 1e-5), n2=_Norm(sd, p + "norms_2.%d" % i, device, 1e-5)))

    def __call__(self, x, mask):
        """x [B,C,T] (conditioning already added) -> DDSConv(x) * mask."""
        n = len(self.layers)
        for i, L in enumerate(self.layers):
            y = ops.channel_norm(x, _new(x), L["n1"].gamma, L["n1"].beta, L["n1"].eps, dw_w=L["dw_w"], dw_bias=L["dw_b"],
                                 dw_dilation=L["dil"], in_mask=mask, act=ACT_GELU)
            y2 = _new(x)
            ops.conv1d(L["pw"], y, y2)
            x = ops.channel_norm(y2, _new(x), L["n2"].gamma, L["n2"].beta, L["n2"].eps, act=ACT_GELU, post_res=x,
                                 out_mask=mask if i == n - 1 else None)
        return x


def _lang_cond(sd, p, device, has_speaker_cond):
    """`cond_lang` 1x1 (language embedding -> channel offsets); when a speaker `cond` exists too the pair is also packed
    as ONE 1x1 over the concatenated [g ; lang] vector, so both offsets cost a single launch."""
    if (p + "cond_lang.weight") not in sd:
        return None, None
    wl, bl = sd[p + "cond_lang.weight"].float(), sd[p + "cond_lang.bias"].float()
    both = None
    if has_speaker_cond:
        both = PackedConv(torch.cat([sd[p + "cond.weight"].float(), wl], 1), sd[p + "cond.bias"].float() + bl, device)
    return PackedConv(wl, bl, device), both


def _cond_offsets(mod, g, lang):
    """Per-(item, channel) conditioning offsets cond(g) + cond_lang(lang) as one [B, C] row_bias operand, or None."""
    use_g = g is not None and mod.cond is not None
    use_l = lang is not None and mod.cond_lang is not None
    if use_g and use_l:
        return ops.speaker_cond(mod.cond_both, torch.cat([g, lang], 1))
    if use_g:
        return ops.speaker_cond(mod.cond, g)
    if use_l:
        return ops.speaker_cond(mod.cond_lang, lang)
    return None


# ------------------------------------------------------------------------------------------------
# StochasticDurationPredictor (reverse) — stochastic_duration_predictor.py:150-294
# ------------------------------------------------------------------------------------------------
class StochasticDurationPredictor:
    def __init__(self, sd, p, device, in_channels, hidden, kernel_size, num_flows=4, cond_channels=0):
        self.hidden, self.num_flows = hidden, num_flows
        self.pre = PackedConv(sd[p + "pre.weight"], sd[p + "pre.bias"], device)
        self.convs = DDSConv(sd, p + "convs.", device, hidden, kernel_size, 3)
        self.proj = PackedConv(sd[p + "proj.weight"], sd[p + "proj.bias"], device)
        self.cond = None
        if cond_channels and (p + "cond.weight") in sd:
            self.cond = PackedConv(sd[p + "cond.weight"], sd[p + "cond.bias"], device)
        self.cond_lang, self.cond_both = _lang_cond(sd, p, device, self.cond is not None)
        self.ea_m = _dev(sd[p + "flows.0.translation"].reshape(-1), device)
        self.ea_logs = _dev(sd[p + "flows.0.log_scale"].reshape(-1), device)
        self.flows = {}
        for i in range(1, num_flows + 1):
            q = p + "flows.%d." % i
            self.flows[i] = dict(
                pre_w=_dev(sd[q + "pre.weight"].reshape(-1), device), pre_b=_dev(sd[q + "pre.bias"], device),
                convs=DDSConv(sd, q + "convs.", device, hidden, kernel_size, 3),
                proj=PackedConv(sd[q + "proj.weight"], sd[q + "proj.bias"], device))
            self.num_bins = (sd[q + "proj.weight"].shape[0] + 1) // 3

    def __call__(self, x, mask, noise, noise_scale=1.0, g=None, lang=None):
        """x [B,C,T] (text-encoder hidden), mask [B,T], noise [B,2,T] -> logw [B,T] view of z[:,0].
        lang [B,L,1]: language embedding, `x + cond_lang(lang_emb)` (:235-236)."""
        h = _new(x, self.hidden)
        row_bias = _cond_offsets(self, g, lang)
        ops.conv1d(self.pre, x, h, row_bias=row_bias)          # pre(x) + cond(g) + cond_lang(l)  (:230-236)
        h = self.convs(h, mask)
        cond = _new(h)
        ops.conv1d(self.proj, h, cond, out_mask=mask)
        order = list(reversed(range(self.num_flows + 1)))
        order = order[:-2] + [order[-1]]           # drop the "useless" flow (:285-286)
        z = ops.scale(noise, noise_scale) if noise_scale != 1.0 else noise.contiguous()
        for i in order:
            z_out = torch.empty_like(z)
            if i == 0:
                ops.sdp_affine_reverse(z_out, z, self.ea_m, self.ea_logs, mask)
            else:
                F_ = self.flows[i]
                hh = _new(cond)
                ops.convflow_pre(hh, z, 1, F_["pre_w"], F_["pre_b"], cond)   # pre(x0) + g, x0 = flip(z)[:,0] = z[:,1]
                hh = F_["convs"](hh, mask)
                par = _new(hh, F_["proj"].c_out)
                ops.conv1d(F_["proj"], hh, par, out_mask=mask)
                ops.convflow_spline_reverse(z_out, z, par, mask, self.num_bins, float(self.hidden), 5.0)
            z = z_out
        return z[:, 0]


# ------------------------------------------------------------------------------------------------
# DurationPredictor — TTS/tts/layers/glow_tts/duration_predictor.py:7-69
# ------------------------------------------------------------------------------------------------
class DurationPredictor:
    def __init__(self, sd, p, device):
        self.cond = PackedConv(sd[p + "cond.weight"], sd[p + "cond.bias"], device) if (p + "cond.weight") in sd else None
        self.cond_lang, self.cond_both = _lang_cond(sd, p, device, self.cond is not None)
        self.c1 = PackedConv(sd[p + "conv_1.weight"], sd[p + "conv_1.bias"], device)
        self.c2 = PackedConv(sd[p + "conv_2.weight"], sd[p + "conv_2.bias"], device)
        self.n1 = _Norm(sd, p + "norm_1", device, 1e-4)
        self.n2 = _Norm(sd, p + "norm_2", device, 1e-4)
        self.proj = PackedConv(sd[p + "proj.weight"], sd[p + "proj.bias"], device)

    def __call__(self, x, mask, g=None, lang=None):
        """x [B,C,T] -> logw [B,T]   (conv -> relu -> LayerNorm twice, then 1x1; dropout is off in eval)."""
        rb = _cond_offsets(self, g, lang)
        if rb is not None:
            x = ops.add_row_bias(x, rb)                                 # x + cond(g) + cond_lang(l)  (:58-62)
        h = _new(x, self.c1.c_out)
        ops.conv1d(self.c1, x, h, in_mask=mask, out_act=ACT_RELU)
        h = ops.channel_norm(h, _new(h), self.n1.gamma, self.n1.beta, self.n1.eps)
        h2 = _new(h, self.c2.c_out)
        ops.conv1d(self.c2, h, h2, in_mask=mask, out_act=ACT_RELU)
        h2 = ops.channel_norm(h2, _new(h2), self.n2.gamma, self.n2.beta, self.n2.eps)
        o = _new(h2, 1)
        ops.conv1d(self.proj, h2, o, in_mask=mask, out_mask=mask)
        return o[:, 0]


# ------------------------------------------------------------------------------------------------
# WaveNet block — TTS/tts/layers/generic/wavenet.py:16-123
# ------------------------------------------------------------------------------------------------
class WN:
    def __init__(self, sd, p, device, hidden, kernel_size, dilation_rate, num_layers, cond_channels=0):
        self.hidden, self.num_layers = hidden, num_layers
        self.in_layers, self.rs_layers = [], []
        for i in range(num_layers):
            w, b = ops.gate_permute(fold_weight_norm(sd, p + "in_layers.%d" % i), sd[p + "in_layers.%d.bias" % i], hidden)
            self.in_layers.append(PackedConv(w, b, device, dilation=dilation_rate ** i))
            self.rs_layers.append(PackedConv(fold_weight_norm(sd, p + "res_skip_layers.%d" % i),
                                             sd[p + "res_skip_layers.%d.bias" % i], device))
        self.cond = None
        if cond_channels and (p + "cond_layer.bias") in sd:
            self.cond_w = fold_weight_norm(sd, p + "cond_layer")
            self.cond = PackedConv(self.cond_w, sd[p + "cond_layer.bias"], device)
            idx = ops.pair_index(hidden, hidden)
            if min(idx) < 0:
                raise ops._lib.TtsAmdError("WN speaker conditioning needs hidden_channels %% %d == 0" % ops.PAIR_ROWS)
            self.gate_idx = torch.tensor(idx, device=device)

    def __call__(self, x, mask, out, g=None, out_kw=None):
        """x [B,H,T] is updated IN PLACE layer by layer; `out` receives sum of skips * mask.
        wavenet.py:92-116; the tanh*sigmoid gate (:6-13) lives in the in_layer conv's epilogue."""
        H = self.hidden
        gl = None
        if g is not None and self.cond is not None:   # cond_layer(g) sliced per layer (wavenet.py:98-105), gate row order
            gl = ops.speaker_cond(self.cond, g).reshape(g.shape[0], self.num_layers, 2 * H)[:, :, self.gate_idx].contiguous()
        acts = _new(x)
        for i in range(self.num_layers):
            ops.conv1d(self.in_layers[i], x, acts, mode=CONV_GATE, row_bias=None if gl is None else gl[:, i].contiguous())
            if i < self.num_layers - 1:
                ops.conv1d(self.rs_layers[i], acts, x, mode=CONV_RES_SKIP, res=x, out_mask=mask, y2=out,
                           accum=out if i > 0 else None, split_row=H)
            else:
                ops.conv1d(self.rs_layers[i], acts, out, accum=out if i > 0 else None, out_mask=mask)
        return out


# ------------------------------------------------------------------------------------------------
# ResidualCouplingBlocks (reverse) — TTS/tts/layers/vits/networks.py:103-232
# ------------------------------------------------------------------------------------------------
class ResidualCouplingBlocks:
    """The channel flip before every flow (networks.py:229-231) is folded into the weights: flows
    that see a flipped tensor read their conditioning half through input-channel-reversed `pre`
    weights and write the coupled half through output-row-reversed `post` weights, so the latent
    stays in place in ONE buffer for the whole stack."""

    def __init__(self, sd, p, device, channels, hidden, kernel_size, dilation_rate, num_layers, num_flows=4,
                 cond_channels=0):
        self.half, self.hidden, self.num_flows = channels // 2, hidden, num_flows
        if num_flows % 2:
            raise ops._lib.TtsAmdError("ResidualCouplingBlocks: the folded channel flips need an even number of flows")
        self.flows = []
        for i in range(num_flows):
            q = p + "flows.%d." % i
            flipped = (num_flows - i) % 2 == 1     # number of flips applied before flow i runs (reverse order)
            wpre, bpre = sd[q + "pre.weight"].float(), sd[q + "pre.bias"].float()
            wpost, bpost = sd[q + "post.weight"].float(), sd[q + "post.bias"].float()
            if wpost.shape[0] != self.half:
                raise ops._lib.TtsAmdError("only mean_only=True coupling (VITS default) has a HIP path")
            if flipped:
                wpre = torch.flip(wpre, [1])
                wpost, bpost = torch.flip(wpost, [0]), torch.flip(bpost, [0])
            self.flows.append(dict(flipped=flipped, pre=PackedConv(wpre, bpre, device),
                                   post=PackedConv(wpost, bpost, device),
                                   wn=WN(sd, q + "enc.", device, hidden, kernel_size, dilation_rate, num_layers,
                                         cond_channels)))

    def forward_flow(self, z, mask, g=None):
        """Forward direction (networks.py:221-225; voice conversion, vits.py:1226): x1 = post(WN(pre(x0))) * mask +
        x1 * mask, then flip — IN PLACE on z with the same folded flips as the reverse pass."""
        half = self.half
        h = _new(z, self.hidden)
        out = _new(z, self.hidden)
        for i in range(self.num_flows):
            F_ = self.flows[i]
            src, dst = (half, 0) if F_["flipped"] else (0, half)
            ops.conv1d(F_["pre"], z, h, c_in_offset=src, out_mask=mask)
            F_["wn"](h, mask, out, g=g)
            ops.conv1d(F_["post"], out, z, res=z, res_row_offset=dst, y_row_offset=dst, out_mask=mask)   # (m + x1) * mask
        return z

    def __call__(self, z, mask, g=None):
        """z [B,C,T] is transformed IN PLACE (reverse direction) and returned."""
        half = self.half
        h = _new(z, self.hidden)
        out = _new(z, self.hidden)
        for i in reversed(range(self.num_flows)):
            F_ = self.flows[i]
            src, dst = (half, 0) if F_["flipped"] else (0, half)
            ops.conv1d(F_["pre"], z, h, c_in_offset=src, out_mask=mask)
            F_["wn"](h, mask, out, g=g)
            # x1 = (x1 - post(h) * mask) * mask   (mean_only: exp(-log_scale) == 1), in place on the other half
            ops.conv1d(F_["post"], out, z, mode=CONV_COUPLE, res=z, res_row_offset=dst, y_row_offset=dst, out_mask=mask)
        return z


# ------------------------------------------------------------------------------------------------
# Glow-TTS encoder — TTS/tts/layers/glow_tts/encoder.py:15-179 (rel_pos_transformer type)
# ------------------------------------------------------------------------------------------------
class GlowEncoder:
    def __init__(self, sd, p, device, hidden, out_channels, encoder_params, mean_only=True, use_prenet=True):
        self.hidden, self.out_channels, self.mean_only = hidden, out_channels, mean_only
        self.emb = _dev(sd[p + "emb.weight"], device)
        self.prenet = None
        if use_prenet:  # ResidualConv1dLayerNormBlock(hidden, hidden, hidden, kernel 5, 3 layers), glow.py:11-67
            self.prenet = dict(
                convs=[PackedConv(sd[p + "prenet.conv_layers.%d.weight" % i], sd[p + "prenet.conv_layers.%d.bias" % i],
                                  device) for i in range(3)],
                norms=[_Norm(sd, p + "prenet.norm_layers.%d" % i, device, 1e-4) for i in range(3)],
                proj=PackedConv(sd[p + "prenet.proj.weight"], sd[p + "prenet.proj.bias"], device))
        ep = encoder_params
        self.encoder = RelativePositionTransformer(sd, p + "encoder.", device, ep["num_layers"], ep["num_heads"],
                                                   ep["kernel_size"], ep.get("rel_attn_window_size"),
                                                   ep.get("layer_norm_type", "1"))
        self.proj_m = PackedConv(sd[p + "proj_m.weight"], sd[p + "proj_m.bias"], device)
        self.proj_s = None if mean_only else PackedConv(sd[p + "proj_s.weight"], sd[p + "proj_s.bias"], device)
        self.duration_predictor = DurationPredictor(sd, p + "duration_predictor.", device)

    def __call__(self, tokens, x_mask, g=None):
        """-> o_mean [B,80,T], o_log_scale [B,80,T] or None (mean_only: zeros), logw [B,T]   (encoder.py:143-179).
        g [B,Cg,1]: speaker vector, concatenated over time to the duration predictor's input (:166-168)."""
        B, T = tokens.shape
        x = torch.empty((B, self.hidden, T), dtype=torch.float32, device=tokens.device)
        ops.embed(tokens, self.emb, x_mask, math.sqrt(self.hidden), x)   # masked here; every consumer masks anyway
        if self.prenet is not None:
            h = x
            for conv, nrm in zip(self.prenet["convs"], self.prenet["norms"]):
                c = _new(x)
                ops.conv1d(conv, h, c, in_mask=x_mask, out_mask=x_mask)
                h = ops.channel_norm(c, _new(x), nrm.gamma, nrm.beta, nrm.eps, act=ACT_RELU)
            x2 = _new(x)
            ops.conv1d(self.prenet["proj"], h, x2, res=x, out_mask=x_mask)
            x = x2
        x = self.encoder(x, x_mask)
        o_mean = _new(x, self.out_channels)
        ops.conv1d(self.proj_m, x, o_mean, out_mask=x_mask)
        o_logs = None
        if self.proj_s is not None:
            o_logs = _new(x, self.out_channels)
            ops.conv1d(self.proj_s, x, o_logs, out_mask=x_mask)
        if g is not None:
            # a constant row is NOT a bias here: conv_1 (k=3) sees zeros beyond the sequence ends and in masked columns
            xd = _new(x, x.shape[1] + g.shape[1])
            xd[:, : x.shape[1]].copy_(x)
            xd[:, x.shape[1]:].copy_(g.expand(-1, -1, x.shape[2]))
            x = xd
        logw = self.duration_predictor(x, x_mask)
        return o_mean, o_logs, logw


# ------------------------------------------------------------------------------------------------
# Glow-TTS decoder (reverse) — TTS/tts/layers/glow_tts/decoder.py:50-141, glow.py:70-233
# ------------------------------------------------------------------------------------------------
class GlowDecoder:
    def __init__(self, sd, p, device, in_channels, hidden, kernel_size, dilation_rate, num_flow_blocks,
                 num_coupling_layers, num_splits=4, num_squeeze=2, sigmoid_scale=False, cond_channels=0):
        if sigmoid_scale:
            raise ops._lib.TtsAmdError("GlowDecoder: sigmoid_scale=True has no HIP epilogue (GlowTTSConfig default is False)")
        self.nsq, self.ns = num_squeeze, num_splits
        c = in_channels * num_squeeze
        self.c, self.half, self.hidden = c, c // 2, hidden
        half = self.half
        self.blocks = []
        for b in range(num_flow_blocks):
            pa, pi, pc = (p + "flows.%d." % (3 * b + j) for j in range(3))
            w_end, b_end = sd[pc + "end.weight"].float(), sd[pc + "end.bias"].float()   # [c, hidden, 1]: t rows | s rows
            # paired-row packing for the COUPLE_AFFINE epilogue: 32-row tile m = t rows [16m, 16m+16) then the matching s rows
            wp, bp = ops.pair_permute(w_end, b_end, half, half)
            w_inv = sd[pi + "weight_inv"] if (pi + "weight_inv") in sd else torch.inverse(sd[pi + "weight"].float())
            self.blocks.append(dict(
                start=PackedConv(fold_weight_norm(sd, pc + "start"), sd[pc + "start.bias"], device),
                wn=WN(sd, pc + "wn.", device, hidden, kernel_size, dilation_rate, num_coupling_layers,
                      cond_channels=cond_channels),
                end=PackedConv(wp, bp, device),
                w_inv=_dev(w_inv.reshape(num_splits, num_splits), device),     # store_inverse(), glow.py:139-141
                w_fwd=_dev(sd[pi + "weight"].float().reshape(num_splits, num_splits), device) if (pi + "weight") in sd else None,
                an_bias=_dev(sd[pa + "bias"].reshape(-1), device), an_logs=_dev(sd[pa + "logs"].reshape(-1), device)))
            blk = self.blocks[-1]
            # the reverse block's InvConvNear^-1 + ActNorm^-1 parameters as ONE device block: they ride in the `end` conv's
            # epilogue (CONV_COUPLE_AFFINE_MIX) — a flow block is then 10 launches, not 11
            blk["mix"] = torch.cat([blk["w_inv"].reshape(-1), blk["an_bias"], blk["an_logs"]]).contiguous()
        self.fuse_mix = num_splits == 4 and c % 4 == 0 and self.half % 2 == 0

    def __call__(self, z, y_mask, g=None):
        """z [B,C,T] (masked), y_mask [B,T] -> mel [B,C,T]; reverse pass: for every block (last to first)
        CouplingBlock^-1, InvConvNear^-1, ActNorm^-1, all in place on the squeezed buffer.  g [B,Cg,1] conditions
        every coupling WaveNet (glow.py:199-213)."""
        B, C, T = z.shape
        x, mq = ops.glow_squeeze(z, y_mask, self.nsq)
        h = _new(x, self.hidden)
        out = _new(x, self.hidden)
        for blk in reversed(self.blocks):
            ops.conv1d(blk["start"], x, h, out_mask=mq)                       # start(x0) * mask   (x0 = first half)
            blk["wn"](h, mq, out, g=g)
            if self.fuse_mix:
                ops.conv1d(blk["end"], out, x, mode=CONV_COUPLE_AFFINE_MIX, res=x, res_row_offset=self.half,
                           y_row_offset=self.half, out_mask=mq, split_row=self.half, y2=blk["mix"])
            else:
                ops.conv1d(blk["end"], out, x, mode=CONV_COUPLE_AFFINE, res=x, res_row_offset=self.half,
                           y_row_offset=self.half, out_mask=mq, split_row=self.half)
                ops.glow_invconv_actnorm(x, blk["w_inv"], blk["an_bias"], blk["an_logs"], mq, self.ns)
        return ops.glow_unsqueeze(x, mq, self.nsq, (T // self.nsq) * self.nsq)

    def forward_flow(self, y, y_mask, g=None):
        """mel y [B,C,T] -> latent z [B,C,T'] (forward pass, decoder.py:113-137 with reverse=False; used by
        GlowTTS.inference_with_MAS / decoder_inference): per block ActNorm, InvConvNear (the weight itself), coupling."""
        B, C, T = y.shape
        x, mq = ops.glow_squeeze(y, y_mask, self.nsq)
        h = _new(x, self.hidden)
        out = _new(x, self.hidden)
        for blk in self.blocks:
            if blk["w_fwd"] is None:
                raise ops._lib.TtsAmdError("GlowDecoder.forward_flow needs flows.*.weight (only weight_inv was stored)")
            ops.glow_invconv_actnorm(x, blk["w_fwd"], blk["an_bias"], blk["an_logs"], mq, self.ns, forward=True)
            ops.conv1d(blk["start"], x, h, out_mask=mq)
            blk["wn"](h, mq, out, g=g)
            ops.conv1d(blk["end"], out, x, mode=CONV_COUPLE_AFFINE_FWD, res=x, res_row_offset=self.half,
                       y_row_offset=self.half, out_mask=mq, split_row=self.half)        # z1 = (t + exp(s) * x1) * mask
        return ops.glow_unsqueeze(x, mq, self.nsq, (T // self.nsq) * self.nsq)


# ------------------------------------------------------------------------------------------------
# PosteriorEncoder — TTS/tts/layers/vits/networks.py:235-288 (voice conversion only at inference time)
# ------------------------------------------------------------------------------------------------
class PosteriorEncoder:
    def __init__(self, sd, p, device, hidden, kernel_size, dilation_rate, num_layers, cond_channels=0):
        self.hidden = hidden
        self.pre = PackedConv(sd[p + "pre.weight"], sd[p + "pre.bias"], device)
        self.enc = WN(sd, p + "enc.", device, hidden, kernel_size, dilation_rate, num_layers, cond_channels)
        self.proj = PackedConv(sd[p + "proj.weight"], sd[p + "proj.bias"], device)

    def __call__(self, y, mask, noise, g=None):
        """y [B,C_spec,T], mask [B,T], noise [B,H,T] -> z [B,H,T], stats [B,2H,T] (mean | log_scale)."""
        h = _new(y, self.hidden)
        ops.conv1d(self.pre, y, h, out_mask=mask)
        out = _new(h)
        self.enc(h, mask, out, g=g)
        stats = _new(h, 2 * self.hidden)
        ops.conv1d(self.proj, out, stats, out_mask=mask)
        return ops.sample_gaussian(stats, noise, mask), stats
