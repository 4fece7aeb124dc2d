"""VITS on hand-written HIP kernels — drop-in for the inference surface of
`TTS.tts.models.vits.Vits` (vits.py:544-724 args/wiring, :1088-1173 inference, :1698-1725
load_checkpoint, :1771-1804 init_from_config).

`Synthesizer`/`synthesis()` only touch: `init_from_config`, `load_checkpoint(config, path, eval=True)`,
`.cuda()`, `parameters()`, `.tokenizer`, `.ap`, `.speaker_manager`, `.language_manager` and
`inference(x, aux_input) -> {"model_outputs", "alignments", ...}`; all are here with
the reference's names and argument meaning.  Training, ONNX export, voice conversion are out of scope.
"""
import os

import torch

from . import _lib, graphs, helpers, layers, ops, parallel
from .hifigan import HifiganGenerator

VITS_ARGS_DEFAULTS = dict(  # VitsArgs, vits.py:544-600
    num_chars=100, out_channels=513, spec_segment_size=32, hidden_channels=192, hidden_channels_ffn_text_encoder=768,
    num_heads_text_encoder=2, num_layers_text_encoder=6, kernel_size_text_encoder=3, dropout_p_text_encoder=0.1,
    dropout_p_duration_predictor=0.5, kernel_size_posterior_encoder=5, dilation_rate_posterior_encoder=1,
    num_layers_posterior_encoder=16, kernel_size_flow=5, dilation_rate_flow=1, num_layers_flow=4,
    resblock_type_decoder="1", resblock_kernel_sizes_decoder=[3, 7, 11],
    resblock_dilation_sizes_decoder=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], upsample_rates_decoder=[8, 8, 2, 2],
    upsample_initial_channel_decoder=512, upsample_kernel_sizes_decoder=[16, 16, 4, 4], use_sdp=True,
    noise_scale=1.0, inference_noise_scale=0.667, length_scale=1.0, noise_scale_dp=1.0, inference_noise_scale_dp=1.0,
    max_inference_len=None, init_discriminator=True, use_speaker_embedding=False, num_speakers=0,
    use_d_vector_file=False, d_vector_dim=0, speaker_embedding_channels=256, condition_dp_on_speaker=True,
    use_language_embedding=False, embedded_language_dim=4, num_languages=0, encoder_sample_rate=None,
    interpolate_z=True,
)


def _get(cfg, name, default=None):
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


class _Args(dict):
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


def _pad_cols(d, T):
    """[B, T0] -> [B, T] zero-extended."""
    out = torch.zeros((d.shape[0], T), dtype=d.dtype, device=d.device)
    out[:, : d.shape[1]] = d
    return out


class Vits:
    def __init__(self, config=None, ap=None, tokenizer=None, speaker_manager=None, language_manager=None):
        self.config = config
        margs = _get(config, "model_args", config)          # VitsConfig.model_args (vits_config.py) or a flat dict
        self.args = _Args(VITS_ARGS_DEFAULTS)
        for k in VITS_ARGS_DEFAULTS:
            v = _get(margs, k, None)
            if v is not None:
                self.args[k] = v
        self.ap, self.tokenizer = ap, tokenizer
        self.speaker_manager, self.language_manager = speaker_manager, language_manager
        a = self.args
        self.length_scale = a.length_scale
        self.inference_noise_scale = a.inference_noise_scale
        self.inference_noise_scale_dp = a.inference_noise_scale_dp
        self.max_inference_len = a.max_inference_len
        # init_multilingual (vits.py:783-803).  The reference takes num_languages from the LanguageManager; without one
        # (checkpoint-only deployment) args.num_languages / the emb_l table in the checkpoint decide.
        self.embedded_language_dim = 0
        if a.use_language_embedding and (language_manager or a.num_languages > 0):
            self.embedded_language_dim = a.embedded_language_dim
        self.emb_l = None
        # init_multispeaker (vits.py:729-778): learned table or external d-vectors
        self.embedded_speaker_dim = 0
        num_speakers = speaker_manager.num_speakers if speaker_manager else a.num_speakers      # vits.py:741-745
        if a.use_speaker_embedding and num_speakers > 0:
            self.embedded_speaker_dim = a.speaker_embedding_channels
        if a.use_d_vector_file:
            if self.embedded_speaker_dim:
                raise ValueError("[!] Speaker embedding layer already initialized before d_vector settings.")
            self.embedded_speaker_dim = a.d_vector_dim
        self.emb_g = None
        # encoder at a lower sample rate than the decoder (vits.py:806-812): the latent is interpolated before decoding
        self.interpolate_factor = None
        if a.encoder_sample_rate:
            sr = _get(_get(config, "audio", None), "sample_rate", None)
            if sr is None:
                raise ValueError("encoder_sample_rate needs config.audio.sample_rate")
            self.interpolate_factor = sr / a.encoder_sample_rate
        self.waveform_decoder = HifiganGenerator(
            a.hidden_channels, 1, a.resblock_type_decoder, a.resblock_dilation_sizes_decoder,
            a.resblock_kernel_sizes_decoder, a.upsample_kernel_sizes_decoder, a.upsample_initial_channel_decoder,
            a.upsample_rates_decoder, inference_padding=0, cond_channels=self.embedded_speaker_dim,
            conv_pre_weight_norm=False,
            conv_post_weight_norm=False, conv_post_bias=False)  # vits.py:704-718
        self.device = torch.device("cpu")
        self._sd = None
        self.text_encoder = self.duration_predictor = self.flow = None
        # the text encoder + duration predictor (~170 launches of a few microseconds) replay as one hipGraph per input
        # shape; set `use_graphs = False` for eager launches
        self.use_graphs = True
        self._front = graphs.GraphCache(self._front_eager)
        self._scratch = graphs.StreamScratch()       # per-stream fixed buffers the graphs read in place (see inference)
        # Small requests (the reference's own call pattern is ONE sentence at a time, synthesizer.py:384) are launch-bound
        # end to end: everything after the one host sync (prior expansion, flows, waveform decoder: ~110 launches on three
        # streams) replays as a second hipGraph.  The decoder length is padded to a multiple of 32 frames so that requests
        # share captures, and runs ragged-exact (every conv treats the row as ending at its own length), which reproduces
        # the unpadded run bit for bit (as long as padding does not move a launch across the small-grid threshold of
        # ttsamd_conv1d_set_small_grid, where the fp32 summation order changes).
        self._tail = graphs.GraphCache(self._tail_eager, max_entries=12)
        self._scratch.dependents += [self._front, self._tail]      # evicting a scratch set drops the graphs that read it in place
        self.weights_version = 0                                   # bumped by every re-pack
        self._tail_cfg = None
        self.graph_tail_max_frames = 2048      # B * padded frames up to which the tail is captured
        self.text_bucket = 16                  # token-axis padding of graphed requests (1 = off): 16 lengths share a capture
        # Plain requests of a single-speaker model run behind the model-level C handle (include/tts_amd.h: ttsamd_vits_*,
        # csrc/vits_model.hip; tts_amd/native.py): the whole launch sequence, the duration sync and the front end's graph replay are
        # C++ — this class then only marshals pointers and draws the two noise tensors.  The handle gets the weights THIS class
        # folded, so it is bitwise the Python-driven path below, which stays for what the handle's envelope leaves out (speaker /
        # language conditioning, ragged-exact batches, latent interpolation).  Single requests replay front end and tail as
        # hipGraphs inside the handle (same-box A/B against this class's own `_front` / `_tail` graphs: 3.32-3.36 vs 3.23-3.35 ms,
        # profiles/r06_native_ab.txt); TTSAMD_NATIVE_SINGLE=0 keeps them on the Python host.  TTSAMD_NATIVE_MODELS=0 /
        # use_native = False: Python-driven everywhere.
        self.use_native = os.environ.get("TTSAMD_NATIVE_MODELS", "1") != "0"
        self.native_single_requests = os.environ.get("TTSAMD_NATIVE_SINGLE", "1") != "0"
        self._native = {}                      # stream handle -> NativeVits (a handle holds ONE request's state)
        self._native_sd = None

    # ---- plug-in surface ---------------------------------------------------------------------------
    @staticmethod
    def init_from_config(config, samples=None, verbose=True):  # vits.py:1771-1804 (no dataset-driven managers here)
        up = _get(_get(config, "model_args", config), "upsample_rates_decoder", VITS_ARGS_DEFAULTS["upsample_rates_decoder"])
        hop = _get(_get(config, "audio", None), "hop_length", None)
        if hop is not None:
            prod = 1
            for u in up:
                prod *= u
            assert prod == hop, " [!] Product of upsample rates must be equal to the hop length - %d vs %d" % (prod, hop)
        return Vits(config, ap=_get(config, "_ap", None), tokenizer=_get(config, "_tokenizer", None),
                    speaker_manager=_get(config, "_speaker_manager", None),
                    language_manager=_get(config, "_language_manager", None))

    def parameters(self):
        return iter([self.text_encoder.emb] if self.text_encoder is not None else [])

    def eval(self):
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def to(self, device):
        self.device = torch.device(device)
        if self._sd is not None:
            self._pack()
        return self

    def load_state_dict(self, sd, strict=True):
        self._sd = {k: v.detach().cpu() for k, v in sd.items() if not k.startswith("disc.")}   # discriminator: training only
        if self.device.type == "cuda":
            self._pack()

    def load_checkpoint(self, config, checkpoint_path, eval=False, strict=True, cache=False):  # noqa: A002
        """vits.py:1698-1725: `{"model": state_dict}`; weight-norm stays parametrised in the checkpoint and is
        folded once here (the reference re-normalises on every forward)."""
        state = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        self.load_state_dict(state["model"], strict=strict)

    def _pack(self):
        if self.device.type != "cuda":
            raise _lib.TtsAmdError("tts_amd.Vits runs only on a GPU (no CPU fallback)")
        a, sd, dev = self.args, self._sd, self.device
        # captured graphs hold raw pointers to the weight tensors replaced below: drop them first
        self._front.clear()
        self._tail.clear()
        self.weights_version += 1
        self._drop_native()
        self.text_encoder = layers.TextEncoder(sd, "text_encoder.", dev, a.hidden_channels, a.num_layers_text_encoder,
                                               a.num_heads_text_encoder, a.kernel_size_text_encoder)
        spk = self.embedded_speaker_dim
        self.emb_g = sd["emb_g.weight"].to(dev, torch.float32).contiguous() if (spk and "emb_g.weight" in sd) else None
        self.emb_l = None
        if self.embedded_language_dim:
            if "emb_l.weight" not in sd:
                raise _lib.TtsAmdError("use_language_embedding is set but the checkpoint has no emb_l.weight")
            self.emb_l = sd["emb_l.weight"].to(dev, torch.float32).contiguous()
        if a.use_sdp:
            self.duration_predictor = layers.StochasticDurationPredictor(sd, "duration_predictor.", dev, a.hidden_channels,
                                                                         192, 3, 4, cond_channels=spk)
        else:
            self.duration_predictor = layers.DurationPredictor(sd, "duration_predictor.", dev)
        self.flow = layers.ResidualCouplingBlocks(sd, "flow.", dev, a.hidden_channels, a.hidden_channels, a.kernel_size_flow,
                                                  a.dilation_rate_flow, a.num_layers_flow, cond_channels=spk)
        self.waveform_decoder.load_state_dict(sd, prefix="waveform_decoder.")
        self.waveform_decoder.to(dev)
        self.posterior_encoder = None
        if "posterior_encoder.pre.weight" in sd:   # kept only for voice conversion (vits.py:1202-1228)
            self.posterior_encoder = layers.PosteriorEncoder(sd, "posterior_encoder.", dev, a.hidden_channels,
                                                             a.kernel_size_posterior_encoder,
                                                             a.dilation_rate_posterior_encoder,
                                                             a.num_layers_posterior_encoder, cond_channels=spk)

    def weight_bytes(self):
        return sum(v.numel() * 4 for v in self._sd.values())

    # ---- the model-level C handle (tts_amd/native.py) ---------------------------------------------------
    def _drop_native(self):
        for n in self._native.values():
            n.close()
        self._native, self._native_sd = {}, None

    def _native_for_stream(self):
        """The NativeVits of the current stream (one per request lane: a handle holds one request's state), built on first use from
        the weights folded here; rebuilt when the inference scales were changed on the object."""
        from . import native

        key = torch.cuda.current_stream().cuda_stream
        nat = self._native.get(key)
        scales = (float(self.inference_noise_scale), float(self.inference_noise_scale_dp), float(self.length_scale))
        if nat is not None and tuple(round(v, 6) for v in nat.scales) != tuple(round(v, 6) for v in scales):
            nat.close()
            nat = None
        if nat is None:
            if self._native_sd is None:
                sd = {}
                for k, v in self._sd.items():
                    if k.startswith("posterior_encoder.") or k.startswith("emb_"):
                        continue
                    if k.endswith(".parametrizations.weight.original0") or k.endswith(".weight_g"):
                        name = k[: -len(".parametrizations.weight.original0")] if k.endswith("original0") else k[: -len(".weight_g")]
                        sd[name + ".weight"] = ops.fold_weight_norm(self._sd, name)
                    elif not (k.endswith(".parametrizations.weight.original1") or k.endswith(".weight_v")):
                        sd[k] = v
                self._native_sd = sd
            if len(self._native) >= 4:           # lanes come and go: keep the handles of the four most recent streams
                self._native.pop(next(iter(self._native))).close()
            nat = self._native[key] = native.NativeVits(self, self._native_sd)
        return nat

    def _native_inference(self, x, x_lengths, durations, a_in, no_graph):
        """A plain request behind the model-level C handle (ttsamd_vits_encode / _decode).  With graphs on, everything the captured
        front end reads — ids, lengths, the duration predictor's noise — is staged into per-stream buffers at fixed addresses (the
        handle keys its captures on them) in ONE launch, and the token axis is padded to the text bucket so that 16 lengths share a
        capture (pad ids masked out by x_lengths own no frames: the valid positions see the unpadded run)."""
        dev = x.device
        B, T0 = x.shape
        nat = self._native_for_stream()
        cb = self.waveform_decoder.concurrent_branches
        nat.set_concurrent_branches((parallel.active_lanes() <= 1) if cb == "auto" else bool(cb))
        graphing = bool(self.use_graphs) and not no_graph
        run_dp = durations is None or bool(a_in.get("run_duration_predictor"))
        noise_dp = a_in.get("noise_dp") if (run_dp and self.args.use_sdp) else None
        T = T0
        if graphing:
            if run_dp and self.text_bucket > 1 and T0 % self.text_bucket:
                T = -(-T0 // self.text_bucket) * self.text_bucket
            sc = self._scratch.get(("native", B, T), lambda: dict(
                x=torch.zeros((B, T), dtype=torch.int64, device=dev), xl=torch.empty((B,), dtype=torch.int64, device=dev),
                nd=torch.zeros((B, 2, T), dtype=torch.float32, device=dev)))
            dst, src = [sc["x"][:, :T0], sc["xl"]], [x, x_lengths.to(dev, torch.int64)]
            if run_dp and self.args.use_sdp:
                # drawn at the reference's shape [B, 2, T0] (a fixed seed gives the same draw, bucketed or not)
                if noise_dp is None and T == T0:
                    torch.randn((B, 2, T0), device=dev, dtype=torch.float32, out=sc["nd"])
                else:
                    dst.append(sc["nd"][:, :, :T0])
                    src.append(torch.randn(B, 2, T0, device=dev, dtype=torch.float32) if noise_dp is None else noise_dp.to(dev, torch.float32))
                noise_dp = sc["nd"]
            ops.copy_into(dst, src)
            x, x_lengths = sc["x"], sc["xl"]
        d = None
        if durations is not None:
            d = durations.to(dev, torch.float32).reshape(B, T0)
            d = d if T == T0 else _pad_cols(d, T)
        t_dec, _ = nat.encode(x, x_lengths, noise_dp, d, bool(a_in.get("run_duration_predictor")), use_graph=graphing)
        extras = bool(a_in.get("return_extras"))
        if T != T0 and B == 1 and graphing and t_dec <= 2048 - 31:      # (the handle's own bound on a replayed tail)
            return nat.decode(t_dec, a_in.get("noise_z"), extras=extras, use_graph=True, t_text_out=T0)      # cut in the handle's copy-out
        return self._cut_text(nat.decode(t_dec, a_in.get("noise_z"), extras=extras, use_graph=graphing), T0, T)

    def _native_request(self, aux_input, B):
        """Does this request run behind the handle?  (See `use_native`.)"""
        from . import native

        a = aux_input or {}
        if not (self.use_native and native.vits_in_envelope(self) and self.args.num_layers_flow > 0):
            return False
        if any(a.get(k) is not None for k in ("speaker_ids", "d_vectors", "language_ids")) or a.get("ragged_exact"):
            return False
        if B == 1 and self.use_graphs and not a.get("no_graph") and not self.native_single_requests:
            return False          # (opt-out) single requests on the Python host's own captured front + tail
        return True

    def _front_eager(self, x, x_mask, noise_dp, g_dp, lang):
        """Text encoder + duration predictor: tokens -> (hidden, prior stats, logw).  g_dp [B,C,1] / lang [B,L,1] or
        empty tensors."""
        g = g_dp if g_dp.numel() else None
        lang = lang if lang.numel() else None
        h, stats = self.text_encoder(x, x_mask, lang=None if lang is None else lang[:, :, 0])
        if self.args.use_sdp:
            logw = self.duration_predictor(h, x_mask, noise_dp, self.inference_noise_scale_dp, g=g, lang=lang)
        else:
            logw = self.duration_predictor(h, x_mask, g=g, lang=lang)
        return h, stats, logw.contiguous()

    def _tail_eager(self, stats, cum, x_mask, y_lengths, noise_z, g):
        """Everything after the output extent is known (vits.py:1152-1161) at the padded length `self._tail_cfg[0]`:
        prior expansion, alignment path, flows, waveform decoder (ragged-exact).  noise_z [B,C,t_pad] is drawn by the
        caller (outside any capture); g: tensor or empty tensor."""
        t_pad, noise_scale = self._tail_cfg
        B, H = stats.shape[0], self.args.hidden_channels
        g = g if g.numel() else None
        # noise_z: the packed [B, C, max(y_lengths)] draw at the head of the [B, C, t_pad] scratch buffer
        pri = ops.expand_prior(stats[:, :H], stats[:, H:], noise_z, cum, x_mask, y_lengths, t_pad, noise_scale, second_copy=True,
                               noise_packed=True)
        attn = ops.generate_path(cum, x_mask, y_lengths, t_pad)
        z = self.flow(pri["z_p2"], pri["y_mask"], g=g)
        # (z * y_mask: the decoder's own stage-0 length mask IS y_mask — sequence_mask(y_lengths) — so no separate in_mask)
        o = self.waveform_decoder.forward(z, g=g, lengths=y_lengths)
        return o, attn, z, pri["z_p"], pri["m_p"], pri["logs_p"], pri["y_mask"]

    def _speaker_g(self, aux_input, B, dev):
        """_set_cond_input / _set_speaker_input (vits.py:873-905,1112-1117): g [B, C_spk, 1] or None."""
        aux_input = aux_input or {}
        sid, dvec = aux_input.get("speaker_ids"), aux_input.get("d_vectors")
        if sid is None and dvec is None:
            return None
        if not self.embedded_speaker_dim:
            raise ValueError("[!] speaker_ids / d_vectors given to a single-speaker model.")
        if dvec is not None and sid is not None:
            raise ValueError("[!] Cannot use d-vectors and speaker-ids together.")
        if dvec is not None:
            dvec = dvec.to(dev, torch.float32)
            return ops.l2_normalize(dvec.reshape(-1, dvec.shape[-1])).unsqueeze(-1)
        if self.emb_g is None:
            raise ValueError("[!] Cannot use speaker-ids without enabling speaker embedding.")
        sid = sid.to(dev, torch.int64).reshape(-1, 1).contiguous()
        g = torch.empty((sid.shape[0], self.emb_g.shape[1], 1), dtype=torch.float32, device=dev)
        return ops.embed(sid, self.emb_g, None, 1.0, g)               # emb_g(sid).unsqueeze(-1)

    @torch.no_grad()
    def voice_conversion(self, y, y_lengths, speaker_cond_src, speaker_cond_tgt, noise=None):
        """vits.py:1202-1228.  y [B, C_spec, T] linear spectrogram (the reference's `wav_to_spec` STFT front end is
        training-side DSP, not built: pass the spectrogram), speaker conds = ids (use_speaker_embedding) or d-vectors.
        Returns (o_hat [B,1,T*hop], y_mask [B,1,T], (z, z_p, z_hat))."""
        if not self.embedded_speaker_dim:
            raise RuntimeError(" [!] Voice conversion is only supported on multi-speaker models.")
        if self.posterior_encoder is None:
            raise _lib.TtsAmdError("voice_conversion needs the posterior_encoder.* weights in the checkpoint")
        _lib.require_gpu(y, "y")
        dev = y.device
        y = y.float().contiguous()
        B, _, T = y.shape
        key = "d_vectors" if self.args.use_d_vector_file else "speaker_ids"
        to_t = lambda v: v if torch.is_tensor(v) else torch.as_tensor(v)   # noqa: E731
        g_src = self._speaker_g({key: to_t(speaker_cond_src)}, B, dev)
        g_tgt = self._speaker_g({key: to_t(speaker_cond_tgt)}, B, dev)
        mask = ops.sequence_mask(y_lengths.to(dev), T)
        H = self.args.hidden_channels
        if noise is None:
            noise = torch.randn(B, H, T, device=dev, dtype=torch.float32)
        z, _ = self.posterior_encoder(y, mask, noise.to(dev, torch.float32), g=g_src)
        z_p = self.flow.forward_flow(z.clone(), mask, g=g_src)
        z_hat = self.flow(z_p.clone(), mask, g=g_tgt)
        o_hat = self.waveform_decoder.forward(z_hat, g=g_tgt, in_mask=mask)
        return o_hat, mask.unsqueeze(1), (z, z_p, z_hat)

    # ---- alignment (vits.py:909-936, 1018-1031) --------------------------------------------------------
    @torch.no_grad()
    def forward_mas(self, outputs, z_p, m_p, logs_p, x, x_mask, y_mask, g=None, lang_emb=None):
        """vits.py:909-942, the alignment half: log-likelihood matrix (MFMA kernel, stays on the device), monotonic
        alignment search (bit-exact HIP kernels; the reference copies logp to the host and runs Cython), durations of the
        path.  x_mask [B,1,T_x] or [B,T_x], y_mask likewise.  Returns (outputs, attn [B,1,T_x,T_y]) like the reference;
        `outputs` gains "attn_durations" [B,1,T_x] (= attn.sum(3), :922).  The duration-predictor LOSS of :921-941
        (incl. its log-duration target) is training code and is not computed; x, g, lang_emb are accepted for signature
        parity and unused."""
        attn = helpers.mas_attention(z_p, m_p, logs_p, x_mask, y_mask)                 # [B, T_x, T_y]
        dur = ops.row_sum(attn).unsqueeze(1)                                            # attn.sum(3) -> [B, 1, T_x]
        outputs["attn_durations"] = dur
        return outputs, attn.unsqueeze(1)

    @torch.no_grad()
    def align(self, x, x_lengths, y, y_lengths, aux_input=None, noise=None):
        """The alignment pass of Vits.forward (vits.py:1018-1031) without its losses: text encoder -> posterior encoder ->
        flow (forward) -> forward_mas -> prior statistics expanded along the path.  x int64 [B,T_x], y [B,C_spec,T_y]
        linear spectrogram (the STFT front end is training-side DSP: pass the spectrogram), `noise` [B,C,T_y] pins the
        posterior's randn_like draw.  -> dict(alignments [B,1,T_x,T_y], attn_durations, z, z_p, m_p, logs_p (expanded),
        x_mask, y_mask)."""
        if self.text_encoder is None or self.posterior_encoder is None:
            raise _lib.TtsAmdError("Vits.align needs text_encoder.* and posterior_encoder.* weights on the GPU")
        _lib.require_gpu(x, "x")
        dev = x.device
        x = x.to(torch.int64).contiguous()
        y = y.to(dev, torch.float32).contiguous()
        B, Tx = x.shape
        Ty = y.shape[2]
        g = self._speaker_g(aux_input, B, dev)
        lang = self._language_emb(aux_input, B, dev)
        x_mask = ops.sequence_mask(x_lengths.to(dev), Tx)
        y_mask = ops.sequence_mask(y_lengths.to(dev), Ty)
        h, stats = self.text_encoder(x, x_mask, lang=None if lang is None else lang[:, :, 0])
        H = self.args.hidden_channels
        m_p, logs_p = stats[:, :H].contiguous(), stats[:, H:].contiguous()
        if noise is None:
            noise = torch.randn(B, H, Ty, device=dev, dtype=torch.float32)
        z, _ = self.posterior_encoder(y, y_mask, noise.to(dev, torch.float32), g=g)
        z_p = self.flow.forward_flow(z.clone(), y_mask, g=g)
        outputs, attn = self.forward_mas({}, z_p, m_p, logs_p, h, x_mask, y_mask, g=g, lang_emb=lang)
        # einsum("klmn, kjm -> kjn", attn, m_p) (vits.py:1029-1030): multiplying by a 0/1 monotonic path is a gather along
        # it — the prior-expansion kernel of the inference path, fed with the path's cumulative durations
        _, cum, ylen = ops.durations(None, x_mask, 1.0, durations_in=outputs["attn_durations"].reshape(B, Tx).contiguous())
        pri = ops.expand_prior(m_p, logs_p, None, cum, x_mask, ylen, Ty, 0.0)
        outputs.update(alignments=attn, z=z, z_p=z_p, x=h, x_mask=x_mask.unsqueeze(1), y_mask=y_mask.unsqueeze(1),
                       m_p=pri["m_p"], logs_p=pri["logs_p"])
        return outputs

    # ---- inference (vits.py:1088-1173) ---------------------------------------------------------------
    def _language_emb(self, aux_input, B, dev):
        """vits.py:886-887,1119-1122: lang_emb = emb_l(language_ids).unsqueeze(-1) -> [B, L, 1] or None."""
        lid = (aux_input or {}).get("language_ids")
        if lid is None or not self.args.use_language_embedding:
            return None
        if self.emb_l is None:
            raise ValueError("[!] language_ids given to a model without a language embedding table.")
        lid = torch.as_tensor(lid).to(dev, torch.int64).reshape(-1, 1).contiguous()
        if lid.shape[0] == 1 and B > 1:
            lid = lid.expand(B, 1).contiguous()
        out = torch.empty((lid.shape[0], self.emb_l.shape[1], 1), dtype=torch.float32, device=dev)
        return ops.embed(lid, self.emb_l, None, 1.0, out)

    @torch.no_grad()
    def inference(self, x, aux_input={"x_lengths": None, "d_vectors": None, "speaker_ids": None,  # noqa: B006
                                      "language_ids": None, "durations": None}):
        """x int64 [B, T_seq]; aux_input["x_lengths"] [B] for batches.  Extra (optional) aux keys let a caller
        pin the two random draws: "noise_dp" [B,2,T_seq] (stochastic_duration_predictor.py:287) and
        "noise_z" [B,C,T_dec] (vits.py:1155); without them they are drawn with torch.randn on the device."""
        if self.text_encoder is None:
            raise _lib.TtsAmdError("tts_amd.Vits: no weights loaded / not moved to the GPU")
        _lib.require_gpu(x, "x")
        a = self.args
        dev = x.device
        x = x.to(torch.int64).contiguous()
        B, T = x.shape
        x_lengths = aux_input.get("x_lengths") if aux_input else None
        if x_lengths is None:
            x_lengths = torch.full((B,), T, dtype=torch.int64, device=dev)       # vits.py:1082-1086
        durations = aux_input.get("durations") if aux_input else None
        no_graph = bool((aux_input or {}).get("no_graph", False))
        if self._native_request(aux_input, B):
            return self._native_inference(x, x_lengths, durations, aux_input or {}, no_graph)
        # Text-length buckets: real traffic brings a new token count with almost every request, and a captured front end is
        # keyed by its shape.  With graphs on, the token axis is padded to a multiple of 16 (pad ids masked out by x_mask —
        # exactly the situation of a shorter sentence inside a batch: masked convs / attention / flows give the valid
        # positions the values of the unpadded run), so 16 lengths share one capture; outputs are cut back at the end.
        T0 = T
        need_dp = durations is None or bool((aux_input or {}).get("run_duration_predictor"))
        graphing = bool(self.use_graphs) and not no_graph
        if graphing and need_dp and self.text_bucket > 1 and T % self.text_bucket:
            T = -(-T // self.text_bucket) * self.text_bucket
        H = a.hidden_channels
        # Everything a graphed request writes before a replay lives in per-stream scratch at fixed addresses (the graphs read
        # it in place): the request's eager launches are then ONE staging copy (ids, pinned noise), the mask, the noise draws,
        # the durations kernel and ONE launch for all output copies — a tensor library issued ~30 small launches here.
        sc = None
        if graphing:
            sc = self._scratch.get((B, T), lambda: dict(
                x=torch.zeros((B, T), dtype=torch.int64, device=dev), x_mask=torch.empty((B, T), dtype=torch.float32, device=dev),
                noise_dp=torch.zeros((B, 2, T), dtype=torch.float32, device=dev),
                dur=torch.empty((B, T), dtype=torch.float32, device=dev), cum=torch.empty((B, T), dtype=torch.int32, device=dev),
                ylen=torch.empty((B,), dtype=torch.int64, device=dev)))
        stage_dst, stage_src = [], []
        if sc is not None:
            # pad ids beyond T0 (left over from an earlier request of the same bucket) are masked out by x_mask — exactly the
            # situation of a shorter sentence inside a batch
            stage_dst.append(sc["x"][:, :T0])
            stage_src.append(x)
            x = sc["x"]
        elif T != T0:
            xp = torch.zeros((B, T), dtype=torch.int64, device=dev)
            xp[:, :T0] = x
            x = xp
        x_mask = ops.sequence_mask(x_lengths.to(dev), T, out=None if sc is None else sc["x_mask"])
        g = self._speaker_g(aux_input, B, dev)
        g_dp = g if a.condition_dp_on_speaker else None
        lang = self._language_emb(aux_input, B, dev)
        empty = torch.empty(0, device=dev)
        # the reference skips the duration predictor when durations are injected (vits.py:1124-1143);
        # "run_duration_predictor" keeps it in the pass anyway (bench.py: fixed output length, no work skipped)
        logw = None
        if need_dp:
            noise_dp = aux_input.get("noise_dp") if aux_input else None
            if not a.use_sdp:
                noise_dp = empty
            elif sc is not None:
                # drawn at the reference's shape [B, 2, T0] (a fixed seed gives the same draw, bucketed or not); columns of the
                # bucket beyond T0 keep whatever an earlier request left there: masked positions, as in a batch
                if noise_dp is None and T == T0:
                    torch.randn((B, 2, T0), device=dev, dtype=torch.float32, out=sc["noise_dp"])
                else:
                    nd = torch.randn(B, 2, T0, device=dev, dtype=torch.float32) if noise_dp is None else \
                        noise_dp.to(dev, torch.float32)
                    stage_dst.append(sc["noise_dp"][:, :, :T0])
                    stage_src.append(nd)
                noise_dp = sc["noise_dp"]
            else:
                if noise_dp is None:
                    noise_dp = torch.randn(B, 2, T0, device=dev, dtype=torch.float32)
                noise_dp = noise_dp.to(dev, torch.float32).contiguous()
                if T != T0:
                    nd = torch.zeros((B, 2, T), dtype=torch.float32, device=dev)
                    nd[:, :, :T0] = noise_dp
                    noise_dp = nd
            if stage_dst:
                ops.copy_into(stage_dst, stage_src)
            gd = g_dp if g_dp is not None else empty
            self._front.enabled = graphing
            ld = lang if lang is not None else empty
            h, stats, logw = self._front(x, x_mask, noise_dp, gd, ld, key=float(self.inference_noise_scale_dp),
                                         stable=(0, 1, 2) if (sc is not None and a.use_sdp) else ((0, 1) if sc is not None else ()))
        else:
            if stage_dst:
                ops.copy_into(stage_dst, stage_src)
            h, stats = self.text_encoder(x, x_mask, lang=None if lang is None else lang[:, :, 0])
        dout = None if sc is None else (sc["dur"], sc["cum"], sc["ylen"])
        if durations is None:
            w_ceil, cum, y_lengths, t_dec = ops.durations(logw.contiguous(), x_mask, float(self.length_scale), want_max=True,
                                                          out=dout)
        else:
            d = durations.to(dev, torch.float32).reshape(B, T0).contiguous()      # vits.py:1141-1143 (+ batches)
            w_ceil, cum, y_lengths, t_dec = ops.durations(None, x_mask, 1.0, durations_in=d if T == T0 else _pad_cols(d, T),
                                                          want_max=True, out=dout)
        # t_dec = max(y_lengths): the request's one host wait (the output extent), polled from a pinned mirror
        noise_z = aux_input.get("noise_z") if aux_input else None
        ragged = bool(aux_input.get("ragged_exact")) if aux_input else False
        t_pad = -(-t_dec // 32) * 32
        if (graphing and sc is not None and (B == 1 or ragged) and self.interpolate_factor is None
                and self.max_inference_len is None and B * t_pad <= self.graph_tail_max_frames):
            # The draw happens OUTSIDE the captured segment, at the reference's shape [B, C, t_dec] (randn_like(m_p),
            # vits.py:1155): with a fixed torch seed the audio is then the same whether the tail replays as a graph, runs
            # eagerly, or was captured earlier (a draw inside the segment would be [B, C, t_pad] and the capture's warm-up
            # runs would advance the generator).  It lands packed at the head of the tail's fixed noise buffer; the kernel
            # reads columns beyond t_dec as zero (masked there anyway).
            nzb = self._scratch.get(("nz", B, t_pad), lambda: torch.zeros(B * H * t_pad, dtype=torch.float32, device=dev))
            packed = nzb[: B * H * t_dec].view(B, H, t_dec)
            if noise_z is None:
                torch.randn((B, H, t_dec), device=dev, dtype=torch.float32, out=packed)
            else:
                assert tuple(noise_z.shape) == (B, H, t_dec), "noise_z must be [B, C, T_dec]"
                ops.copy_into([packed], [noise_z.to(dev, torch.float32)])
            self._tail.enabled = True
            self._tail_cfg = (t_pad, float(self.inference_noise_scale))
            o, attn, z, z_p, m_p, logs_p, y_mask = self._tail(
                stats, cum, x_mask, y_lengths, nzb.view(B, H, t_pad), g if g is not None else empty,
                key=self._tail_cfg, stable=(0, 1, 2, 3, 4) if (need_dp and self._front.last_static) else (1, 2, 3, 4))
            hop = o.shape[-1] // t_pad
            # the graph's outputs are static buffers (overwritten by its next replay): hand out copies, cut to the true
            # extent (and to the caller's T0 tokens) — ONE launch for all of them
            extras = bool(aux_input and aux_input.get("return_extras"))
            views = [o[:, :, : t_dec * hop], attn[:, :T0, :t_dec], w_ceil[:, :T0].unsqueeze(1), z[:, :, :t_dec], z_p[:, :, :t_dec],
                     m_p[:, :, :t_dec], logs_p[:, :, :t_dec], y_mask[:, :t_dec].unsqueeze(1),
                     y_lengths if (ragged or extras) else None, h[:, :, :T0] if extras else None,
                     logw[:, :T0].unsqueeze(1) if (extras and logw is not None) else None]
            c = ops.clone_views(views)
            outputs = {"model_outputs": c[0], "alignments": c[1], "durations": c[2], "z": c[3], "z_p": c[4], "m_p": c[5],
                       "logs_p": c[6], "y_mask": c[7]}
            if ragged or extras:
                outputs["y_lengths"] = c[8]
            if extras:
                outputs.update(x=c[9], logw=c[10])
            return outputs
        if sc is not None:      # the eager tail hands its tensors out: they must not alias the per-stream scratch
            w_ceil, cum, y_lengths, x_mask = ops.clone_views([w_ceil, cum, y_lengths, x_mask])
        if noise_z is None:
            noise_z = torch.randn(B, H, t_dec, device=dev, dtype=torch.float32)
        noise_z = noise_z.to(dev, torch.float32).contiguous()
        assert noise_z.shape == (B, H, t_dec), "noise_z must be [B, C, T_dec]"
        pri = ops.expand_prior(stats[:, :H], stats[:, H:], noise_z, cum, x_mask, y_lengths, t_dec,
                               float(self.inference_noise_scale), second_copy=True)
        attn = ops.generate_path(cum, x_mask, y_lengths, t_dec)
        y_mask = pri["y_mask"]
        z = self.flow(pri["z_p2"], y_mask, g=g)
        dec_lengths = y_lengths
        if self.interpolate_factor is not None and a.interpolate_z:           # upsampling_z, vits.py:944-959
            z = ops.linear_interp(z, self.interpolate_factor)
            dec_lengths = torch.ceil(y_lengths.to(torch.float64) * self.interpolate_factor).to(torch.int64)
            if int(dec_lengths.max().item()) != z.shape[2]:
                raise _lib.TtsAmdError("interpolate_z: y_lengths * interpolate_factor does not match the interpolated "
                                       "latent length (the reference fails on this shape mismatch too)")
            y_mask = ops.sequence_mask(dec_lengths, z.shape[2])
        zd = z if self.max_inference_len is None else z[:, :, : self.max_inference_len].contiguous()
        md = y_mask if self.max_inference_len is None else y_mask[:, : self.max_inference_len].contiguous()
        # "ragged_exact": every decoder conv treats row b as ending at y_lengths[b] -> row b equals a B=1 run (bitwise
        # when both runs take the same conv tile family, fp32 reassociation otherwise: see HifiganGenerator.forward)
        # (z*y_mask)[:, :, :max_len]; ragged: the decoder's stage-0 length mask equals md (sequence_mask(dec_lengths) cut at max_len)
        o = self.waveform_decoder.forward(zd, g=g, in_mask=None if ragged else md, lengths=dec_lengths if ragged else None)
        outputs = {
            "model_outputs": o,
            "alignments": attn,
            "durations": w_ceil.unsqueeze(1),
            "z": z,
            "z_p": pri["z_p"],
            "m_p": pri["m_p"],
            "logs_p": pri["logs_p"],
            "y_mask": y_mask.unsqueeze(1),
        }
        if ragged:
            outputs["y_lengths"] = dec_lengths
        if aux_input and aux_input.get("return_extras"):
            # (h / logw may alias the captured front end's static buffers: hand out copies)
            outputs.update(x=h.clone(), logw=None if logw is None else logw.clone().unsqueeze(1), y_lengths=y_lengths)
        return self._cut_text(outputs, T0, T)

    @staticmethod
    def _cut_text(outputs, T0, T):
        """Undo the text-length bucket: token-indexed outputs back to the caller's T0 tokens."""
        if T != T0:
            outputs["alignments"] = outputs["alignments"][:, :T0].contiguous()
            outputs["durations"] = outputs["durations"][:, :, :T0].contiguous()
            for k in ("x", "logw"):
                if outputs.get(k) is not None:
                    outputs[k] = outputs[k][:, :, :T0].contiguous()
        return outputs

    __call__ = inference
