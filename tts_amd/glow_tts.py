"""Glow-TTS on hand-written HIP kernels — drop-in for the inference surface of
`TTS.tts.models.glow_tts.GlowTTS` (glow_tts.py:59-105 wiring, :341-374 inference, :519-530 store_inverse /
load_checkpoint, :541-557 init_from_config).  Config field names/defaults: GlowTTSConfig
(TTS/tts/configs/glow_tts_config.py:101-152).  Single- and multi-speaker (speaker-embedding table or d-vectors);
training is out of scope.
"""
import os

import torch

from . import _lib, graphs, helpers, layers, ops
from .vits import _Args, _get

GLOW_DEFAULTS = dict(  # glow_tts_config.py:101-152
    num_chars=None, encoder_type="rel_pos_transformer",
    encoder_params=dict(kernel_size=3, dropout_p=0.1, num_layers=6, num_heads=2, hidden_channels_ffn=768,
                        input_length=None),
    use_encoder_prenet=True, hidden_channels_enc=192, hidden_channels_dec=192, hidden_channels_dp=256,
    dropout_p_dp=0.1, dropout_p_dec=0.05, mean_only=True, out_channels=80, num_flow_blocks_dec=12,
    inference_noise_scale=0.0, kernel_size_dec=5, dilation_rate=1, num_block_layers=4, num_speakers=0,
    c_in_channels=0, num_splits=4, num_squeeze=2, sigmoid_scale=False, d_vector_dim=0, length_scale=1.0,
    use_speaker_embedding=False, use_d_vector_file=False,
)


class GlowTTS:
    def __init__(self, config=None, ap=None, tokenizer=None, speaker_manager=None):
        self.config = config
        self.args = _Args(GLOW_DEFAULTS)
        for k in GLOW_DEFAULTS:
            v = _get(config, k, None)
            if v is not None:
                self.args[k] = v
        a = self.args
        for k, v in a.items():      # "pass all config fields to self" (glow_tts.py:69-72)
            setattr(self, k, v)
        self.ap, self.tokenizer, self.speaker_manager, self.language_manager = ap, tokenizer, speaker_manager, None
        self.decoder_output_dim = a.out_channels
        if a.encoder_type != "rel_pos_transformer":
            raise _lib.TtsAmdError("tts_amd.GlowTTS: only encoder_type='rel_pos_transformer' (the config default) is built")
        # init_multispeaker (glow_tts.py:107-135)
        self.embedded_speaker_dim = 0
        if speaker_manager is not None:
            self.num_speakers = speaker_manager.num_speakers
        if a.use_d_vector_file:
            self.embedded_speaker_dim = a.d_vector_dim if a.d_vector_dim else 512
            if speaker_manager is not None and speaker_manager.embedding_dim:
                assert a.d_vector_dim == speaker_manager.embedding_dim, \
                    " [!] d-vector dimension mismatch b/w config and speaker manager."
        self.has_emb_g = bool(a.use_speaker_embedding and not a.use_d_vector_file)
        if self.has_emb_g:
            self.embedded_speaker_dim = a.hidden_channels_enc
        self.c_in_channels = self.embedded_speaker_dim
        self.emb_g = None
        if a.num_chars is None and tokenizer is not None:
            a.num_chars = tokenizer.characters.num_chars
        self.device = torch.device("cpu")
        self._sd = None
        self.encoder = self.decoder = None
        # A sentence is ~230 launches of a few microseconds (encoder ~60, 12 flow blocks ~170): issued one by one the HOST is
        # the bottleneck (~10 us per launch through the C ABI).  As in tts_amd.Vits the encoder + duration predictor replay
        # as one hipGraph per input shape and, for single sentences / ragged-exact batches, everything after the one host
        # sync (prior expansion, alignment path, decoder flows) as a second one at the frame count padded to 32 (masks make
        # the padded run equal the unpadded one).  `use_graphs = False` restores eager launches.
        self.use_graphs = True
        self._front = graphs.GraphCache(self._front_eager)
        self._tail = graphs.GraphCache(self._tail_eager, max_entries=12)
        self._tail_cfg = None
        # per-stream fixed buffers the graphs read in place (see tts_amd.Vits.inference); evicting a set drops the graphs over it
        self._scratch = graphs.StreamScratch(dependents=[self._front, self._tail])
        self.weights_version = 0               # bumped by every re-pack: dependants (SentencePipeline) key their graphs on it
        self.graph_tail_max_frames = 4096      # B * padded frames up to which the tail is captured
        self.text_bucket = 16                  # token-axis padding of graphed requests (1 = off)
        # plain requests of a single-speaker model run behind the model-level C handle (ttsamd_glowtts_*, csrc/glow_model.hip;
        # see tts_amd.Vits.use_native): batches there, single sentences on the captured front + tail of this class
        self.use_native = os.environ.get("TTSAMD_NATIVE_MODELS", "1") != "0"
        self.native_single_requests = os.environ.get("TTSAMD_NATIVE_SINGLE", "0") != "0"
        self._native = {}
        self._native_sd = None

    @staticmethod
    def init_from_config(config, samples=None, verbose=True):
        return GlowTTS(config, ap=_get(config, "_ap", None), tokenizer=_get(config, "_tokenizer", None),
                       speaker_manager=_get(config, "_speaker_manager", None))

    def parameters(self):
        return iter([self.encoder.emb] if self.encoder is not None else [])

    def eval(self):
        return self

    def store_inverse(self):  # done at pack time (4x4 inverses + weight-norm folding)
        pass

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def to(self, device):
        self.device = torch.device(device)
        if self._sd is not None:
            self._pack()
        return self

    def load_state_dict(self, sd, strict=True):
        self._sd = {k: v.detach().cpu() for k, v in sd.items()}
        if self.device.type == "cuda":
            self._pack()

    def load_checkpoint(self, config, checkpoint_path, eval=False):  # noqa: A002
        state = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        self.load_state_dict(state["model"])

    def _pack(self):
        if self.device.type != "cuda":
            raise _lib.TtsAmdError("tts_amd.GlowTTS runs only on a GPU (no CPU fallback)")
        a, sd, dev = self.args, self._sd, self.device
        self._front.clear()          # captured graphs hold raw pointers to the weight tensors replaced below
        self._tail.clear()
        self.weights_version += 1
        self._drop_native()
        self.encoder = layers.GlowEncoder(sd, "encoder.", dev, a.hidden_channels_enc, a.out_channels, a.encoder_params,
                                          a.mean_only, a.use_encoder_prenet)
        self.decoder = layers.GlowDecoder(sd, "decoder.", dev, a.out_channels, a.hidden_channels_dec, a.kernel_size_dec,
                                          a.dilation_rate, a.num_flow_blocks_dec, a.num_block_layers, a.num_splits,
                                          a.num_squeeze, a.sigmoid_scale, cond_channels=self.c_in_channels)
        self.emb_g = None
        if self.has_emb_g:
            if "emb_g.weight" not in sd:
                raise _lib.TtsAmdError("use_speaker_embedding is set but the checkpoint has no emb_g.weight")
            self.emb_g = sd["emb_g.weight"].to(dev, torch.float32).contiguous()

    # ---- the model-level C handle (tts_amd/native.py) ---------------------------------------------------
    def _drop_native(self):
        for n in self._native.values():
            n.close()
        self._native, self._native_sd = {}, None

    def _native_for_stream(self):
        """The NativeGlowTTS of the current stream (a handle holds one request's state), built on first use from the weights folded
        here and the 4 x 4 inverses computed here (store_inverse), so that it is bitwise the Python-driven path."""
        from . import native

        key = torch.cuda.current_stream().cuda_stream
        nat = self._native.get(key)
        scales = (float(self.inference_noise_scale), float(self.length_scale))
        if nat is not None and tuple(round(v, 6) for v in nat.scales) != tuple(round(v, 6) for v in scales):
            nat.close()
            nat = None
        if nat is None:
            if self._native_sd is None:
                sd = {}
                for k, v in self._sd.items():
                    if k.startswith("emb_g"):
                        continue
                    if k.endswith(".parametrizations.weight.original0") or k.endswith(".weight_g"):
                        name = k[: -len(".parametrizations.weight.original0")] if k.endswith("original0") else k[: -len(".weight_g")]
                        sd[name + ".weight"] = ops.fold_weight_norm(self._sd, name)
                    elif not (k.endswith(".parametrizations.weight.original1") or k.endswith(".weight_v")):
                        sd[k] = v
                for k in list(sd):          # InvConvNear: the inverse this class would compute (layers.GlowDecoder)
                    if k.startswith("decoder.flows.") and k.endswith(".weight") and sd[k].dim() == 2 and (k[: -len("weight")] + "weight_inv") not in sd:
                        sd[k[: -len("weight")] + "weight_inv"] = torch.inverse(sd[k].float())
                self._native_sd = sd
            if len(self._native) >= 4:
                self._native.pop(next(iter(self._native))).close()
            nat = self._native[key] = native.NativeGlowTTS(self, self._native_sd)
        return nat

    def _native_request(self, aux_input, B):
        from . import native

        a = aux_input or {}
        if not (self.use_native and native.glow_in_envelope(self)):
            return False
        if any(a.get(k) is not None for k in ("speaker_ids", "d_vectors", "_front_ctx")):
            return False
        if B == 1 and self.use_graphs and not a.get("no_graph") and not self.native_single_requests:
            return False
        return True

    def _speaker_embedding(self, aux_input, dev):
        """glow_tts.py:162-191: g = normalize(emb_g(speaker_ids)) or normalize(d_vectors), [B, C, 1]; None if neither."""
        aux_input = aux_input or {}
        sid, dvec = aux_input.get("speaker_ids"), aux_input.get("d_vectors")
        if dvec is not None and sid is not None:
            raise ValueError("[!] Cannot use d-vectors and speaker-ids together.")
        if sid is not None and self.emb_g is None:
            raise ValueError("[!] Cannot use speaker-ids without enabling speaker embedding.")
        if sid is None and dvec is None:
            return None
        if sid is not None:
            sid = torch.as_tensor(sid).to(dev, torch.int64).reshape(-1, 1).contiguous()
            v = torch.empty((sid.shape[0], self.emb_g.shape[1], 1), dtype=torch.float32, device=dev)
            v = ops.embed(sid, self.emb_g, None, 1.0, v)[:, :, 0]
        else:
            if not self.c_in_channels:
                raise ValueError("[!] d_vectors given to a single-speaker model.")
            v = dvec.to(dev, torch.float32).reshape(-1, dvec.shape[-1])
        return ops.l2_normalize(v.contiguous()).unsqueeze(-1)

    def _front_eager(self, x, x_mask, g):
        """encoder + duration predictor: tokens -> (o_mean, o_logs or empty, logw).  g [B,C,1] or an empty tensor."""
        o_mean, o_logs, logw = self.encoder(x, x_mask, g=g if g.numel() else None)
        return o_mean, (o_logs if o_logs is not None else torch.empty(0, device=x.device)), logw.contiguous()

    def _tail_eager(self, o_mean, o_logs, cum, x_mask, y_lengths, noise, g):
        """everything after the output extent is known, at the padded length self._tail_cfg[0] (glow_tts.py:361-366).
        noise: the packed [B, C, max(y_lengths)] draw at the head of a [B, C, t_pad] scratch buffer, or empty."""
        t_pad, noise_scale = self._tail_cfg
        pri = ops.expand_prior(o_mean, o_logs if o_logs.numel() else None, noise if noise.numel() else None, cum, x_mask,
                               y_lengths, t_pad, noise_scale, mask_out=True, noise_packed=True)
        attn = ops.generate_path(cum, x_mask, y_lengths, t_pad)
        y = self.decoder(pri["z_p"], pri["y_mask"], g=g if g.numel() else None)
        logs_p = pri["logs_p"] if pri["logs_p"] is not None else torch.empty(0, device=o_mean.device)
        return y, attn, pri["m_p"], logs_p, ops.attn_durations(cum, x_mask, y_lengths)

    def request_front(self, x, aux_input=None):
        """First half of `inference` — encoder + duration predictor (one graph replay), durations, and the request's one
        host wait (the output extent) — as a context dict the second half (`inference` itself, or a Synthesizer pipeline that
        continues straight into the vocoder) picks up."""
        if self.encoder is None:
            raise _lib.TtsAmdError("tts_amd.GlowTTS: no weights loaded / not moved to the GPU")
        _lib.require_gpu(x, "x")
        aux_input = aux_input or {}
        dev = x.device
        x = x.to(torch.int64).contiguous()
        B, T = x.shape
        x_lengths = aux_input.get("x_lengths")
        if x_lengths is None:
            x_lengths = torch.full((B,), T, dtype=torch.int64, device=dev)
        no_graph = bool(aux_input.get("no_graph", False))
        graphing = bool(self.use_graphs) and not no_graph
        # text-length buckets (see tts_amd.Vits.inference): the token axis of a graphed request is padded to a multiple of 16,
        # pad ids masked out and owning no frames, so the valid positions see the unpadded run
        # (single sentences and ragged-exact batches only: in a plain batch the reference's one-frame-per-padded-token rule
        # applies to the batch's own padding and is kept as it is)
        T0 = T
        ragged = bool(aux_input.get("ragged_exact"))
        if graphing and (B == 1 or ragged) and self.text_bucket > 1 and T % self.text_bucket:
            T = -(-T // self.text_bucket) * self.text_bucket
        sc = None
        if graphing:       # per-stream scratch at fixed addresses: the graphs read it in place (tts_amd.Vits.inference)
            sc = self._scratch.get((B, T), lambda: dict(
                x=torch.zeros((B, T), dtype=torch.int64, device=dev), x_mask=torch.empty((B, T), dtype=torch.float32, device=dev),
                dur=torch.empty((B, T), dtype=torch.float32, device=dev), cum=torch.empty((B, T), dtype=torch.int32, device=dev),
                ylen=torch.empty((B,), dtype=torch.int64, device=dev)))
            ops.copy_into([sc["x"][:, :T0]], [x])
            x = sc["x"]
        elif T != T0:
            xp = torch.zeros((B, T), dtype=torch.int64, device=dev)
            xp[:, :T0] = x
            x = xp
        x_mask = ops.sequence_mask(x_lengths.to(dev), T, out=None if sc is None else sc["x_mask"])
        g = self._speaker_embedding(aux_input, dev)
        empty = torch.empty(0, device=dev)
        self._front.enabled = graphing
        o_mean, o_logs, logw = self._front(x, x_mask, g if g is not None else empty, stable=(0, 1) if sc is not None else ())
        front_static = self._front.last_static
        o_logs = o_logs if o_logs.numel() else None
        d_in = aux_input.get("durations")
        dout = None if sc is None else (sc["dur"], sc["cum"], sc["ylen"])
        host = {}
        if d_in is not None:   # not a reference feature: lets a parity harness pin the integer durations (ceil cliff)
            d = d_in.to(dev, torch.float32).reshape(B, T0).contiguous()
            if T != T0:
                dp = torch.zeros((B, T), dtype=torch.float32, device=dev)
                dp[:, :T0] = d
                d = dp
            w_ceil, cum, y_lengths, t_dec = ops.durations(None, x_mask, 1.0, durations_in=d, want_max=True, out=dout, host_out=host)
        else:
            # the bucket's own padding (columns >= T0) owns no frames; inside the caller's tensor the reference's rule holds
            # — clamp_min gives every token one frame, masked or not (glow_tts.py:350-351) — unless the caller asked for
            # ragged-exact batching
            w_ceil, cum, y_lengths, t_dec = ops.durations(logw.contiguous(), x_mask, float(self.length_scale),
                                                          glow=2 if ragged else 1, t_valid=T0, want_max=True, out=dout,
                                                          host_out=host)
        return dict(B=B, T=T, T0=T0, dev=dev, sc=sc, graphing=graphing, ragged=ragged, x_mask=x_mask, g=g, o_mean=o_mean,
                    o_logs=o_logs, logw=logw, w_ceil=w_ceil, cum=cum, y_lengths=y_lengths, t_dec=t_dec,
                    y_lengths_host=host["y_lengths"], front_static=front_static)

    def tail_inputs(self, ctx, aux_input=None):
        """The inputs of `_tail_eager` for a request context at its 32-frame bucket -> (t_pad, inputs, stable indices):
        the noise draw (outside any capture, at the reference's shape [B, C, t_dec]) lands packed at the head of the bucket's
        fixed noise buffer."""
        a, dev, B, t_dec = self.args, ctx["dev"], ctx["B"], ctx["t_dec"]
        C = a.out_channels
        empty = torch.empty(0, device=dev)
        noise = (aux_input or {}).get("noise")
        t_pad = -(-t_dec // 32) * 32
        nz = empty
        if noise is not None or self.inference_noise_scale != 0.0:
            nzb = self._scratch.get(("nz", B, t_pad), lambda: torch.zeros(B * C * t_pad, dtype=torch.float32, device=dev))
            packed = nzb[: B * C * t_dec].view(B, C, t_dec)
            if noise is None:
                torch.randn((B, C, t_dec), device=dev, dtype=torch.float32, out=packed)
            else:
                ops.copy_into([packed], [noise.to(dev, torch.float32)])
            nz = nzb.view(B, C, t_pad)
        o_logs, g = ctx["o_logs"], ctx["g"]
        inputs = (ctx["o_mean"], o_logs if o_logs is not None else empty, ctx["cum"], ctx["x_mask"], ctx["y_lengths"], nz,
                  g if g is not None else empty)
        stable = ((0, 1) if ctx["front_static"] else ()) + (2, 3, 4) + ((5,) if nz.numel() else ())
        return t_pad, inputs, stable

    @torch.no_grad()
    def inference(self, x, aux_input={"x_lengths": None, "d_vectors": None, "speaker_ids": None}):  # noqa: B006
        """glow_tts.py:341-374.  Optional aux keys: "noise" [B,C,T_dec] pins the randn_like(y_mean) draw;
        "ragged_exact": padded tokens own no frames (the reference gives each PADDED token one frame via clamp_min,
        which only matters in batches) so that row b equals a B=1 run on sentence b (up to the conv launcher's batch-dependent
        tile choice: fp32 reassociation, ~1e-6 relative)."""
        # "_front_ctx": the context of a request_front() the caller has already run for this very request (a Synthesizer whose
        # fused pipeline declined a large batch): the front end — and the request's host wait — is not repeated
        if self.encoder is not None and x.is_cuda and self._native_request(aux_input, x.shape[0]):
            a_in = aux_input or {}
            nat = self._native_for_stream()
            dev = x.device
            B, T = x.shape
            xl = a_in.get("x_lengths")
            graphing = bool(self.use_graphs) and not a_in.get("no_graph")
            x = x.to(torch.int64)
            if graphing:          # the captured front end reads ids / lengths at fixed addresses: staged per stream in one launch
                sc = self._scratch.get(("native", B, T), lambda: dict(x=torch.zeros((B, T), dtype=torch.int64, device=dev),
                                                                      xl=torch.empty((B,), dtype=torch.int64, device=dev)))
                xl = torch.full((B,), T, dtype=torch.int64, device=dev) if xl is None else xl.to(dev, torch.int64)
                ops.copy_into([sc["x"], sc["xl"]], [x.contiguous(), xl.contiguous()])
                x, xl = sc["x"], sc["xl"]
            t_dec, _ = nat.encode(x, xl, a_in.get("durations"), bool(a_in.get("ragged_exact")), use_graph=graphing)
            return nat.decode(t_dec, a_in.get("noise"))
        ctx = (aux_input or {}).get("_front_ctx") or self.request_front(x, aux_input)
        if (ctx["B"], ctx["T0"]) != tuple(x.shape):
            raise _lib.TtsAmdError("GlowTTS.inference: the front-end context handed in belongs to another request "
                                   "([%d, %d] tokens, this one has %s)" % (ctx["B"], ctx["T0"], tuple(x.shape)))
        a = self.args
        B, T, T0, dev, t_dec, ragged = ctx["B"], ctx["T"], ctx["T0"], ctx["dev"], ctx["t_dec"], ctx["ragged"]
        o_mean, o_logs, logw, w_ceil, cum, y_lengths, x_mask, g = (ctx[k] for k in (
            "o_mean", "o_logs", "logw", "w_ceil", "cum", "y_lengths", "x_mask", "g"))
        t_pad = -(-t_dec // 32) * 32
        if ctx["graphing"] and (B == 1 or ragged) and B * t_pad <= self.graph_tail_max_frames:
            t_pad, inputs, stable = self.tail_inputs(ctx, aux_input)
            self._tail.enabled = True
            self._tail_cfg = (t_pad, float(self.inference_noise_scale))
            y, attn, m_p, logs_p, tot = self._tail(*inputs, key=self._tail_cfg, stable=stable)
            # static buffers of the graph (overwritten by its next replay): hand out copies cut to the true extent (the decoder's
            # squeeze drops the frames that do not fill a group: its output is (t_dec // num_squeeze) * num_squeeze long) and to
            # the caller's T0 tokens — ONE launch for all of them
            t_y = (t_dec // self.num_squeeze) * self.num_squeeze
            c = ops.clone_views([y[:, :, :t_y].transpose(1, 2), m_p[:, :, :t_dec].transpose(1, 2),
                                 logs_p[:, :, :t_dec].transpose(1, 2) if logs_p.numel() else None,
                                 attn[:, :T0, :t_dec].permute(0, 2, 1), logw[:, :T0].unsqueeze(2), tot[:, :T0].unsqueeze(2),
                                 y_lengths, w_ceil[:, :T0].unsqueeze(1)])
            return {"model_outputs": c[0], "logdet": None, "y_mean": c[1], "y_log_scale": c[2], "alignments": c[3],
                    "durations_log": c[4], "total_durations_log": c[5], "y_lengths": c[6], "durations": c[7]}
        noise = aux_input.get("noise") if aux_input else None
        C = a.out_channels
        if noise is None and self.inference_noise_scale != 0.0:
            noise = torch.randn(B, C, t_dec, device=dev, dtype=torch.float32)
        if noise is not None:
            noise = noise.to(dev, torch.float32).contiguous()
        if ctx["sc"] is not None:      # the eager tail hands its tensors out: they must not alias the per-stream scratch
            w_ceil, cum, y_lengths, x_mask = ops.clone_views([w_ceil, cum, y_lengths, x_mask])
        pri = ops.expand_prior(o_mean, o_logs, noise, cum, x_mask, y_lengths, t_dec, float(self.inference_noise_scale),
                               mask_out=True)
        attn = ops.generate_path(cum, x_mask, y_lengths, t_dec)
        y = self.decoder(pri["z_p"], pri["y_mask"], g=g)
        y_log_scale = pri["logs_p"]
        return self._cut_text({
            "model_outputs": y.transpose(1, 2),
            "logdet": None,
            "y_mean": pri["m_p"].transpose(1, 2),
            "y_log_scale": y_log_scale.transpose(1, 2),
            "alignments": attn.permute(0, 2, 1),
            "durations_log": logw.clone().unsqueeze(1).transpose(1, 2),      # (may alias the front graph's static buffer)
            "total_durations_log": ops.attn_durations(cum, x_mask, y_lengths).unsqueeze(1).transpose(1, 2),
            "y_lengths": y_lengths,
            "durations": w_ceil.unsqueeze(1),
        }, T0, T)

    @staticmethod
    def _cut_text(out, T0, T):
        """Undo the text-length bucket: token-indexed outputs back to the caller's T0 tokens."""
        if T != T0:
            out["alignments"] = out["alignments"][:, :, :T0].contiguous()              # [B, T_dec, T_x]
            out["durations"] = out["durations"][:, :, :T0].contiguous()                # [B, 1, T_x]
            for k in ("durations_log", "total_durations_log"):                         # [B, T_x, 1]
                out[k] = out[k][:, :T0].contiguous()
        return out

    __call__ = inference

    def _preprocess(self, y, y_lengths):
        """glow_tts.py:510-517: drop the frames that do not fill a squeeze group."""
        n = self.num_squeeze
        t = (y.shape[2] // n) * n
        return y[:, :, :t].contiguous(), torch.div(y_lengths, n, rounding_mode="floor") * n

    @torch.no_grad()
    def decoder_inference(self, y, y_lengths=None, aux_input=None):
        """glow_tts.py:318-339: mel [B,T,C] -> decoder forward -> decoder reverse (round trip)."""
        _lib.require_gpu(y, "y")
        y = y.float().transpose(1, 2).contiguous()
        if y_lengths is None:
            y_lengths = torch.full((y.shape[0],), y.shape[2], dtype=torch.int64, device=y.device)
        y_mask = ops.sequence_mask(y_lengths.to(y.device), y.shape[2])
        g = self._speaker_embedding(aux_input, y.device)
        z = self.decoder.forward_flow(y, y_mask, g=g)
        out = self.decoder(z, y_mask[:, : z.shape[2]].contiguous(), g=g)
        return {"model_outputs": out.transpose(1, 2), "logdet": None}

    @torch.no_grad()
    def inference_with_MAS(self, x, x_lengths, y=None, y_lengths=None, aux_input=None):
        """glow_tts.py:262-316 ("teacher forcing"): encoder -> decoder FORWARD on the given mel -> log-likelihood matrix
        -> monotonic alignment search (all on the device, no host round trip) -> aligned prior statistics."""
        _lib.require_gpu(x, "x")
        dev = x.device
        x = x.to(torch.int64).contiguous()
        B, T = x.shape
        x_mask = ops.sequence_mask(x_lengths.to(dev), T)
        g = self._speaker_embedding(aux_input, dev)
        o_mean, o_logs, logw = self.encoder(x, x_mask, g=g)
        y = y.float().transpose(1, 2).contiguous()
        y, y_lengths = self._preprocess(y, y_lengths.to(dev))
        y_mask = ops.sequence_mask(y_lengths, y.shape[2])
        z = self.decoder.forward_flow(y, y_mask, g=g)
        zeros = torch.zeros_like(o_mean) if o_logs is None else o_logs        # mean_only: o_log_scale == 0
        attn = helpers.mas_attention(z, o_mean, zeros, x_mask, y_mask, glow_order=True)          # [B, T_x, T_y]
        dur = ops.row_sum(attn)                                                                    # attn.sum(-1)
        _, cum, y_len2 = ops.durations(None, x_mask, 1.0, durations_in=dur)
        t_dec = y.shape[2]
        pri = ops.expand_prior(o_mean, o_logs, None, cum, x_mask, y_lengths, t_dec, 0.0, mask_out=True)
        out = self.decoder(pri["z_p"], y_mask, g=g)      # the reference also decodes the aligned prior (result unused there)
        total = ops.attn_durations(cum, x_mask, y_lengths)       # log(1 + attn.sum(-1)) * x_mask
        return {
            "model_outputs": pri["z_p"].transpose(1, 2),          # z = y_mean * y_mask (glow_tts.py:302,307)
            "logdet": None,
            "y_mean": pri["m_p"].transpose(1, 2),
            "y_log_scale": pri["logs_p"].transpose(1, 2),
            "alignments": attn.permute(0, 2, 1),
            "durations_log": logw.unsqueeze(1).transpose(1, 2),
            "total_durations_log": total.unsqueeze(1).transpose(1, 2),
            "decoded": out.transpose(1, 2),
        }
