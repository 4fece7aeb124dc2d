"""ctypes views of the model-level C ABI of the two acoustic models (include/tts_amd.h: ttsamd_vits_*, ttsamd_glowtts_*;
csrc/vits_model.hip, csrc/glow_model.hip): weight folding / re-ordering / packing, the launch sequences of `Vits.inference`
(TTS/tts/models/vits.py:1088-1173) and `GlowTTS.inference` (TTS/tts/models/glow_tts.py:341-374), the duration sync through a pinned
mirror and the front end's hipGraph replay all live behind a handle in C++ — the boundary a non-Python host binds.
These classes only marshal pointers, allocate the caller-owned outputs and make the two `torch.randn` draws the reference makes.
`tts_amd.vits.Vits` / `tts_amd.glow_tts.GlowTTS` route their plain requests here (see their `use_native`)."""
import ctypes

import torch

from . import _lib, ops
from .hifigan import HifiganConfig

_PREC = {"h2": 0, "x3": 1, "f32": 2}


class VitsConfig(ctypes.Structure):
    """Mirror of `ttsamd_vits_config`."""

    _fields_ = [("num_chars", ctypes.c_int32), ("hidden_channels", ctypes.c_int32), ("hidden_channels_ffn_text_encoder", ctypes.c_int32),
                ("num_heads_text_encoder", ctypes.c_int32), ("num_layers_text_encoder", ctypes.c_int32),
                ("kernel_size_text_encoder", ctypes.c_int32), ("kernel_size_flow", ctypes.c_int32), ("dilation_rate_flow", ctypes.c_int32),
                ("num_layers_flow", ctypes.c_int32), ("num_flows", ctypes.c_int32), ("use_sdp", ctypes.c_int32),
                ("inference_noise_scale", ctypes.c_float), ("inference_noise_scale_dp", ctypes.c_float), ("length_scale", ctypes.c_float),
                ("decoder", HifiganConfig)]


class VitsOutputs(ctypes.Structure):
    """Mirror of `ttsamd_vits_outputs`."""

    _fields_ = [(n, ctypes.c_void_p) for n in ("wav", "alignments", "durations", "z", "z_p", "m_p", "logs_p", "y_mask", "y_lengths", "logw",
                                               "x_hidden")] + [("t_text_out", ctypes.c_int32)]


class GlowConfig(ctypes.Structure):
    """Mirror of `ttsamd_glowtts_config`."""

    _fields_ = [(n, ctypes.c_int32) for n in ("num_chars", "hidden_channels_enc", "hidden_channels_dec", "hidden_channels_dp", "out_channels",
                                              "encoder_kernel_size", "encoder_num_layers", "encoder_num_heads", "encoder_hidden_channels_ffn",
                                              "encoder_rel_attn_window_size", "encoder_layer_norm_type", "use_encoder_prenet", "mean_only",
                                              "num_flow_blocks_dec", "kernel_size_dec", "dilation_rate", "num_block_layers", "num_splits",
                                              "num_squeeze")] + \
               [("inference_noise_scale", ctypes.c_float), ("length_scale", ctypes.c_float), ("precision", ctypes.c_int32)]


class GlowOutputs(ctypes.Structure):
    """Mirror of `ttsamd_glowtts_outputs`."""

    _fields_ = [(n, ctypes.c_void_p) for n in ("mel", "y_mean", "y_log_scale", "alignments", "durations_log", "total_durations_log", "durations",
                                               "y_lengths")]


def hifigan_config(gen, precision=None):
    """`ttsamd_hifigan_config` of a tts_amd.HifiganGenerator (its constructor arguments)."""
    c = HifiganConfig()
    c.in_channels, c.out_channels, c.resblock_type = gen.in_channels, gen.out_channels, int(gen.resblock_type)
    c.num_kernels = gen.num_kernels
    for j, (k, dil) in enumerate(zip(gen.resblock_kernel_sizes, gen.resblock_dilation_sizes)):
        c.resblock_kernel_sizes[j], c.num_dilations[j] = k, len(dil)
        for d, v in enumerate(dil):
            c.resblock_dilation_sizes[j][d] = v
    c.num_upsamples = gen.num_upsamples
    for i, (u, k) in enumerate(zip(gen.upsample_factors, gen.upsample_kernel_sizes)):
        c.upsample_factors[i], c.upsample_kernel_sizes[i] = u, k
    c.upsample_initial_channel, c.inference_padding = gen.upsample_initial_channel, gen.inference_padding
    c.precision = _PREC[precision or ops.conv_precision()]
    return c


def _load_all(load, handle, sd, what):
    for name, t in sd.items():
        t = t.detach().to("cpu", torch.float32).contiguous()
        if t.dim() == 0 or t.numel() == 0:
            continue
        shape = (ctypes.c_int64 * t.dim())(*t.shape)
        _lib.check(load(handle, name.encode(), ctypes.c_void_p(t.data_ptr()), shape, t.dim()), what)


class _Handle:
    _destroy = None

    def close(self):
        if getattr(self, "_h", None):
            getattr(_lib.lib(), self._destroy)(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def vits_in_envelope(model):
    """True when a tts_amd.Vits model is one the handle covers: single speaker, single language, no latent interpolation, no
    output cut (include/tts_amd.h: ttsamd_vits_config)."""
    return (not model.embedded_speaker_dim and not model.embedded_language_dim and model.interpolate_factor is None
            and model.max_inference_len is None)


def glow_in_envelope(model):
    """True when a tts_amd.GlowTTS model is one the handle covers: single speaker, num_splits 4, no sigmoid_scale."""
    a = model.args
    return not model.c_in_channels and a.num_splits == 4 and not a.sigmoid_scale and a.encoder_type == "rel_pos_transformer"


class NativeVits(_Handle):
    """`ttsamd_vits_{create,load,finalize,encode,decode,destroy}` seen from Python.  `model` is a tts_amd.Vits (its VitsArgs and
    waveform-decoder configuration are read from it), `state_dict` defaults to the one it was given."""

    _destroy = "ttsamd_vits_destroy"

    def __init__(self, model, state_dict=None, precision=None):
        sd = state_dict if state_dict is not None else model._sd
        if sd is None:
            raise _lib.TtsAmdError("NativeVits: no weights")
        a = model.args
        c = vits_config(model, sd, precision)
        self.hidden, self.use_sdp = a.hidden_channels, bool(a.use_sdp)
        self.scales = (c.inference_noise_scale, c.inference_noise_scale_dp, c.length_scale)
        self._h = ctypes.c_void_p()
        L = _lib.lib()
        L.ttsamd_vits_hop_length.restype = ctypes.c_int64
        _lib.check(L.ttsamd_vits_create(ctypes.byref(c), ctypes.byref(self._h)), "vits_create")
        try:
            _load_all(L.ttsamd_vits_load, self._h, sd, "vits_load")
            _lib.check(L.ttsamd_vits_finalize(self._h), "vits_finalize")
        except Exception:
            self.close()
            raise
        self.hop = int(L.ttsamd_vits_hop_length(self._h))

    def set_concurrent_branches(self, on):
        """MRF branch streams of the waveform decoder on / off (TTSAMD_HIFIGAN_OPT_CONCURRENT_BRANCHES): on for a lone request, off
        when the host keeps several requests in flight."""
        on = bool(on)
        if getattr(self, "_concurrent", True) != on:
            _lib.check(_lib.lib().ttsamd_vits_set_option(self._h, 1, int(on)), "vits_set_option")
            self._concurrent = on

    @torch.no_grad()
    def encode(self, x, x_lengths=None, noise_dp=None, durations=None, run_duration_predictor=False, use_graph=False):
        """First half of a request -> (t_dec, y_lengths as a list of ints).  The call returns when the frame counts are on the host."""
        _lib.require_gpu(x, "x")
        dev = x.device
        x = x.to(torch.int64).contiguous()
        B, T = x.shape
        xl = torch.full((B,), T, dtype=torch.int64, device=dev) if x_lengths is None else x_lengths.to(dev, torch.int64).contiguous()
        run_dp = durations is None or run_duration_predictor
        if run_dp and self.use_sdp:
            noise_dp = torch.randn(B, 2, T, device=dev, dtype=torch.float32) if noise_dp is None else noise_dp.to(dev, torch.float32).contiguous()
        d = None if durations is None else durations.to(dev, torch.float32).reshape(B, T).contiguous()
        host = (ctypes.c_int64 * B)()
        t_dec = ctypes.c_int32(0)
        _lib.check(_lib.lib().ttsamd_vits_encode(self._h, _lib.P(x), _lib.P(xl), B, T, _lib.P(noise_dp), _lib.P(d), int(bool(run_duration_predictor)),
                                                 host, ctypes.byref(t_dec), int(bool(use_graph)), _lib.stream_ptr()), "vits_encode")
        self._req = (x, xl, noise_dp, d, B, T)          # the launches are asynchronous: inputs stay referenced until decode
        self._ran_dp = run_dp
        return int(t_dec.value), [int(v) for v in host]

    @torch.no_grad()
    def decode(self, t_dec, noise_z=None, extras=False, use_graph=False, t_text_out=None):
        """Second half -> the dict `Vits.inference` returns (vits.py:1163-1173).  t_text_out: the caller's token count when the
        request ran on a padded token axis (single requests with use_graph: the outputs are cut in the handle's copy-out)."""
        x, xl, noise_dp, d, B, T_run = self._req
        T = T_run if t_text_out is None else int(t_text_out)
        dev, H = x.device, self.hidden
        if noise_z is None:
            noise_z = torch.randn(B, H, t_dec, device=dev, dtype=torch.float32)      # randn_like(m_p), vits.py:1155
        noise_z = noise_z.to(dev, torch.float32).contiguous()
        assert tuple(noise_z.shape) == (B, H, t_dec), "noise_z must be [B, C, T_dec]"
        # every fp32 output is a view of ONE allocation (a request is latency-bound: eight torch.empty calls are 20+ us of host time)
        shapes = [("model_outputs", (B, 1, t_dec * self.hop)), ("alignments", (B, T, t_dec)), ("durations", (B, 1, T)), ("z", (B, H, t_dec)),
                  ("z_p", (B, H, t_dec)), ("m_p", (B, H, t_dec)), ("logs_p", (B, H, t_dec)), ("y_mask", (B, 1, t_dec))]
        if extras:
            shapes.append(("x", (B, H, T)))
            if self._ran_dp:
                shapes.append(("logw", (B, 1, T)))
        sizes = [(-(-(a * b * c) // 64)) * 64 for _, (a, b, c) in shapes]          # 256-byte aligned views
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        out, off = {}, 0
        for (k, shp), n in zip(shapes, sizes):
            out[k] = flat[off: off + shp[0] * shp[1] * shp[2]].view(shp)
            off += n
        o = VitsOutputs()
        o.t_text_out = 0 if T == T_run else T
        o.wav, o.alignments, o.durations = out["model_outputs"].data_ptr(), out["alignments"].data_ptr(), out["durations"].data_ptr()
        o.z, o.z_p, o.m_p, o.logs_p, o.y_mask = (out[k].data_ptr() for k in ("z", "z_p", "m_p", "logs_p", "y_mask"))
        if extras:
            out["y_lengths"] = torch.empty(B, dtype=torch.int64, device=dev)
            o.y_lengths, o.x_hidden = out["y_lengths"].data_ptr(), out["x"].data_ptr()
            if self._ran_dp:
                o.logw = out["logw"].data_ptr()
            else:
                out["logw"] = None
        _lib.check(_lib.lib().ttsamd_vits_decode(self._h, _lib.P(noise_z), ctypes.byref(o), int(bool(use_graph)), _lib.stream_ptr()), "vits_decode")
        self._keep = (self._req, noise_z, out)
        return out

    def inference(self, x, x_lengths=None, noise_dp=None, noise_z=None, durations=None, run_duration_predictor=False, use_graph=False,
                  extras=False):
        t_dec, _ = self.encode(x, x_lengths, noise_dp, durations, run_duration_predictor, use_graph)
        return self.decode(t_dec, noise_z, extras, use_graph)


class NativeGlowTTS(_Handle):
    """`ttsamd_glowtts_{create,load,finalize,encode,decode,destroy}` seen from Python (`model`: a tts_amd.GlowTTS)."""

    _destroy = "ttsamd_glowtts_destroy"

    def __init__(self, model, state_dict=None, precision=None):
        sd = state_dict if state_dict is not None else model._sd
        if sd is None:
            raise _lib.TtsAmdError("NativeGlowTTS: no weights")
        a = model.args
        ep = a.encoder_params
        c = GlowConfig()
        c.num_chars = int(sd["encoder.emb.weight"].shape[0])
        c.hidden_channels_enc, c.hidden_channels_dec, c.hidden_channels_dp, c.out_channels = (a.hidden_channels_enc, a.hidden_channels_dec,
                                                                                                a.hidden_channels_dp, a.out_channels)
        c.encoder_kernel_size, c.encoder_num_layers, c.encoder_num_heads = ep["kernel_size"], ep["num_layers"], ep["num_heads"]
        c.encoder_hidden_channels_ffn = ep["hidden_channels_ffn"]
        c.encoder_rel_attn_window_size = int(ep.get("rel_attn_window_size") or 0)
        c.encoder_layer_norm_type = int(ep.get("layer_norm_type", "1"))
        c.use_encoder_prenet, c.mean_only = int(bool(a.use_encoder_prenet)), int(bool(a.mean_only))
        c.num_flow_blocks_dec, c.kernel_size_dec, c.dilation_rate, c.num_block_layers = (a.num_flow_blocks_dec, a.kernel_size_dec, a.dilation_rate,
                                                                                         a.num_block_layers)
        c.num_splits, c.num_squeeze = a.num_splits, a.num_squeeze
        c.inference_noise_scale, c.length_scale = float(model.inference_noise_scale), float(model.length_scale)
        c.precision = _PREC[precision or ops.conv_precision()]
        self.C, self.nsq, self.noise_scale, self.mean_only = a.out_channels, a.num_squeeze, float(model.inference_noise_scale), bool(a.mean_only)
        self.scales = (c.inference_noise_scale, c.length_scale)
        self._h = ctypes.c_void_p()
        L = _lib.lib()
        _lib.check(L.ttsamd_glowtts_create(ctypes.byref(c), ctypes.byref(self._h)), "glowtts_create")
        try:
            _load_all(L.ttsamd_glowtts_load, self._h, sd, "glowtts_load")
            _lib.check(L.ttsamd_glowtts_finalize(self._h), "glowtts_finalize")
        except Exception:
            self.close()
            raise

    @torch.no_grad()
    def encode(self, x, x_lengths=None, durations=None, ragged_exact=False, use_graph=False):
        _lib.require_gpu(x, "x")
        dev = x.device
        x = x.to(torch.int64).contiguous()
        B, T = x.shape
        xl = torch.full((B,), T, dtype=torch.int64, device=dev) if x_lengths is None else x_lengths.to(dev, torch.int64).contiguous()
        d = None if durations is None else durations.to(dev, torch.float32).reshape(B, T).contiguous()
        host = (ctypes.c_int64 * B)()
        t_dec = ctypes.c_int32(0)
        _lib.check(_lib.lib().ttsamd_glowtts_encode(self._h, _lib.P(x), _lib.P(xl), B, T, _lib.P(d), int(bool(ragged_exact)), host,
                                                    ctypes.byref(t_dec), int(bool(use_graph)), _lib.stream_ptr()), "glowtts_encode")
        self._req = (x, xl, d, B, T)
        return int(t_dec.value), [int(v) for v in host]

    @torch.no_grad()
    def decode(self, t_dec, noise=None):
        """-> the dict `GlowTTS.inference` returns (glow_tts.py:362-374; [B, T, C] views of the channels-first buffers)."""
        x, xl, d, B, T = self._req
        dev, C = x.device, self.C
        if noise is None and self.noise_scale != 0.0:
            noise = torch.randn(B, C, t_dec, device=dev, dtype=torch.float32)
        if noise is not None:
            noise = noise.to(dev, torch.float32).contiguous()
        t_y = (t_dec // self.nsq) * self.nsq
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
        mel, y_mean, y_logs, attn = new(B, C, t_y), new(B, C, t_dec), new(B, C, t_dec), new(B, T, t_dec)
        dlog, tot, dur, ylen = new(B, T), new(B, T), new(B, 1, T), torch.empty(B, dtype=torch.int64, device=dev)
        o = GlowOutputs()
        o.mel, o.y_mean, o.y_log_scale, o.alignments = mel.data_ptr(), y_mean.data_ptr(), y_logs.data_ptr(), attn.data_ptr()
        o.durations_log, o.total_durations_log, o.durations, o.y_lengths = dlog.data_ptr(), tot.data_ptr(), dur.data_ptr(), ylen.data_ptr()
        _lib.check(_lib.lib().ttsamd_glowtts_decode(self._h, _lib.P(noise), ctypes.byref(o), _lib.stream_ptr()), "glowtts_decode")
        out = {"model_outputs": mel.transpose(1, 2), "logdet": None, "y_mean": y_mean.transpose(1, 2), "y_log_scale": y_logs.transpose(1, 2),
               "alignments": attn.permute(0, 2, 1), "durations_log": dlog.unsqueeze(2), "total_durations_log": tot.unsqueeze(2),
               "y_lengths": ylen, "durations": dur}
        self._keep = (self._req, noise, out)
        return out

    def inference(self, x, x_lengths=None, noise=None, durations=None, ragged_exact=False, use_graph=False):
        t_dec, _ = self.encode(x, x_lengths, durations, ragged_exact, use_graph)
        return self.decode(t_dec, noise)


def save_flat_weights(sd, path):
    """A reference-layout state_dict as the flat file a non-Python host reads (tests/native/vits_host.cpp): "TTSAMDW1", u32 n, then
    n x { u32 name_len, name, u32 ndim, i64 dims[ndim], f32 data }."""
    import struct

    items = [(k, v.detach().to("cpu", torch.float32).contiguous()) for k, v in sd.items() if v.dim() >= 1 and v.numel() > 0]
    with open(path, "wb") as f:
        f.write(b"TTSAMDW1" + struct.pack("<I", len(items)))
        for k, v in items:
            kb = k.encode()
            f.write(struct.pack("<I", len(kb)) + kb + struct.pack("<I", v.dim()) + struct.pack("<%dq" % v.dim(), *v.shape))
            f.write(v.numpy().tobytes())


def vits_config(model, state_dict=None, precision=None):
    """`ttsamd_vits_config` of a tts_amd.Vits (what NativeVits passes to ttsamd_vits_create)."""
    sd = state_dict if state_dict is not None else model._sd
    a = model.args
    c = VitsConfig()
    c.num_chars = int(sd["text_encoder.emb.weight"].shape[0])
    c.hidden_channels, c.hidden_channels_ffn_text_encoder = a.hidden_channels, a.hidden_channels_ffn_text_encoder
    c.num_heads_text_encoder, c.num_layers_text_encoder = a.num_heads_text_encoder, a.num_layers_text_encoder
    c.kernel_size_text_encoder = a.kernel_size_text_encoder
    c.kernel_size_flow, c.dilation_rate_flow, c.num_layers_flow, c.num_flows = a.kernel_size_flow, a.dilation_rate_flow, a.num_layers_flow, 4
    c.use_sdp = int(bool(a.use_sdp))
    c.inference_noise_scale = float(model.inference_noise_scale)
    c.inference_noise_scale_dp = float(model.inference_noise_scale_dp)
    c.length_scale = float(model.length_scale)
    c.decoder = hifigan_config(model.waveform_decoder, precision)
    return c
