"""HiFiGAN generator on hand-written HIP kernels — drop-in for
`TTS.vocoder.models.hifigan_generator.HifiganGenerator` (hifigan_generator.py:162-301) and the
`TTS.vocoder.models.gan.GAN` inference wrapper (gan.py:58-66, 229-252).

Same constructor arguments, same `forward(x, g=None)`, `inference(c)` (replicate pad, no crop),
`load_checkpoint(config, path, eval)` and state_dict key layout (weight-norm parametrised or
stripped).  Every launch is a fused conv: leaky-ReLU lives in each conv's prologue, bias /
residual add / MRF accumulate-and-average / tanh in its epilogue; ConvTranspose1d runs as a
polyphase 2-tap conv with a pixel-shuffle epilogue; a whole ResBlock1 iteration
(lrelu -> conv(k,d) -> lrelu -> conv(k,1) -> +x) is ONE launch on the 32/64-channel stages (and the
128-channel stage's k=3 blocks) with its intermediate tensor kept in LDS; conv_post is an HBM-streaming
kernel.  v1: 78 conv launches unfused, 54 with fusion.  No elementwise passes over HBM remain.
"""
import collections
import ctypes
import os

import torch

from . import _lib, graphs, ops, parallel
from .ops import ACT_LRELU, ACT_TANH, CONV_SHUFFLE, PackedConv

LRELU_SLOPE = 0.1  # hifigan_generator.py:11


def _cumprod(xs):
    out, p = [], 1
    for x in xs:
        p *= x
        out.append(p)
    return out


class HifiganConfig(ctypes.Structure):
    """Mirror of `ttsamd_hifigan_config` (include/tts_amd.h)."""

    _fields_ = [("in_channels", ctypes.c_int32), ("out_channels", ctypes.c_int32), ("resblock_type", ctypes.c_int32),
                ("num_kernels", ctypes.c_int32), ("resblock_kernel_sizes", ctypes.c_int32 * 8), ("num_dilations", ctypes.c_int32 * 8),
                ("resblock_dilation_sizes", (ctypes.c_int32 * 8) * 8), ("num_upsamples", ctypes.c_int32),
                ("upsample_factors", ctypes.c_int32 * 12), ("upsample_kernel_sizes", ctypes.c_int32 * 12),
                ("upsample_initial_channel", ctypes.c_int32), ("inference_padding", ctypes.c_int32), ("precision", ctypes.c_int32)]


class NativeHifigan:
    """The model-level C ABI of the vocoder (include/tts_amd.h: ttsamd_hifigan_{create,load,finalize,forward,destroy}) seen from
    Python: weight folding / packing, masks, the launch sequence and its hipGraph replay all live behind the handle in C++
    (csrc/hifigan_model.hip) — the boundary a non-Python host binds; this class only marshals pointers."""

    _PREC = {"h2": 0, "x3": 1, "f32": 2}

    def __init__(self, gen, state_dict=None, precision=None):
        """`gen`: a tts_amd.HifiganGenerator, or — no Python generator needed — a dict of HifiganGenerator's constructor arguments
        (in_channels, out_channels, resblock_type, resblock_dilation_sizes, resblock_kernel_sizes, upsample_kernel_sizes,
        upsample_initial_channel, upsample_factors, inference_padding; hifigan_generator.py:163-178) with `state_dict` given."""
        if isinstance(gen, dict):
            import types

            g = types.SimpleNamespace(**gen)
            g.num_kernels, g.num_upsamples = len(g.resblock_kernel_sizes), len(g.upsample_factors)
            g.inference_padding = gen.get("inference_padding", 5)
            g.cond_channels, g.cond_in_each_up_layer, g._sd = gen.get("cond_channels", 0), gen.get("cond_in_each_up_layer", False), None
            gen = g
        sd = state_dict if state_dict is not None else gen._sd
        if sd is None:
            raise _lib.TtsAmdError("NativeHifigan: no weights")
        if gen.cond_channels > 0 or gen.cond_in_each_up_layer:
            raise _lib.TtsAmdError("NativeHifigan: speaker-conditioned generators run through the kernel-level ABI")
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
        c.precision = self._PREC[precision or ops.conv_precision()]
        self.out_channels = gen.out_channels
        self._h = ctypes.c_void_p()
        L = _lib.lib()
        L.ttsamd_hifigan_output_samples.restype = ctypes.c_int64
        _lib.check(L.ttsamd_hifigan_create(ctypes.byref(c), ctypes.byref(self._h)), "hifigan_create")
        try:
            for name, t in sd.items():
                t = t.detach().to("cpu", torch.float32).contiguous()
                if t.dim() == 0 or t.numel() == 0:
                    continue
                shape = (ctypes.c_int64 * t.dim())(*t.shape)
                _lib.check(L.ttsamd_hifigan_load(self._h, name.encode(), ctypes.c_void_p(t.data_ptr()), shape, t.dim()), "hifigan_load")
            _lib.check(L.ttsamd_hifigan_finalize(self._h), "hifigan_finalize")
        except Exception:
            self.close()
            raise

    def output_samples(self, frames):
        return int(_lib.lib().ttsamd_hifigan_output_samples(self._h, int(frames)))

    def set_concurrent_branches(self, on):
        """MRF branch streams on / off (TTSAMD_HIFIGAN_OPT_CONCURRENT_BRANCHES): on for a lone request, off when the host keeps
        several requests in flight."""
        on = bool(on)
        if getattr(self, "_concurrent", True) != on:
            _lib.check(_lib.lib().ttsamd_hifigan_set_option(self._h, 1, int(on)), "hifigan_set_option")
            self._concurrent = on

    @torch.no_grad()
    def forward(self, mel, lengths=None, use_graph=False, out=None):
        """mel [B, C, T] fp32 on the GPU (+ lengths [B] int64 on the GPU for ragged-exact batching) -> wav [B, 1, samples]."""
        _lib.require_gpu(mel, "mel")
        mel = mel.contiguous().float()
        B, _, T = mel.shape
        wav = out if out is not None else torch.empty((B, self.out_channels, self.output_samples(T)), dtype=torch.float32, device=mel.device)
        ln = None if lengths is None else lengths.to(mel.device, torch.int64).contiguous()
        _lib.check(_lib.lib().ttsamd_hifigan_forward(self._h, _lib.P(mel), B, T, _lib.P(ln), _lib.P(wav), int(bool(use_graph)), _lib.stream_ptr()),
                   "hifigan_forward")
        self._keep = (mel, ln, wav)          # the launches are asynchronous: inputs stay referenced until the next call
        return wav

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().ttsamd_hifigan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HifiganGenerator:
    def __init__(self, in_channels, out_channels, resblock_type, resblock_dilation_sizes, resblock_kernel_sizes,
                 upsample_kernel_sizes, upsample_initial_channel, upsample_factors, inference_padding=5,
                 cond_channels=0, conv_pre_weight_norm=True, conv_post_weight_norm=True, conv_post_bias=True,
                 cond_in_each_up_layer=False):
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.resblock_type = str(resblock_type)
        self.resblock_dilation_sizes = [list(d) for d in resblock_dilation_sizes]
        self.resblock_kernel_sizes = list(resblock_kernel_sizes)
        self.upsample_kernel_sizes = list(upsample_kernel_sizes)
        self.upsample_initial_channel = upsample_initial_channel
        self.upsample_factors = list(upsample_factors)
        self.inference_padding = inference_padding
        self.cond_channels = cond_channels
        # XTTS variant (TTS/tts/layers/xtts/hifigan_decoder.py:183-299): `o = ups[i](o) + conds[i](g)` after every upsample
        self.cond_in_each_up_layer = cond_in_each_up_layer
        self.num_kernels = len(self.resblock_kernel_sizes)
        self.num_upsamples = len(self.upsample_factors)
        for u, k in zip(self.upsample_factors, self.upsample_kernel_sizes):
            if k < u:
                raise _lib.TtsAmdError("ConvTranspose1d with kernel < stride (k=%d u=%d) leaves output samples without any tap: "
                                       "the reference's padding (k - u) // 2 is negative there" % (k, u))
        # any (kernel, stride): ConvTranspose1d runs in polyphase form, a ceil(k / u)-tap Conv1d with a pixel-shuffle epilogue
        # (k = 2u: the tuned 2-tap kernels; everything else: the generic kernel).  With k - u odd the reference's output is one
        # sample longer than T * u per stage: fine for forward(); ragged batching / length buckets then do not apply
        self.exact_hop = all((k - u) % 2 == 0 for u, k in zip(self.upsample_factors, self.upsample_kernel_sizes))
        self.device = torch.device("cpu")
        self._sd = None
        self._packed = None
        # MRF resblocks on separate HIP streams (see forward): True / False, or "auto" = three branch streams for a lone
        # request, one stream when the request runs inside parallel.Lanes with two or more lanes (parallel.active_lanes)
        self.concurrent_branches = "auto"
        # ResBlock1 iterations (lrelu -> conv(k,d) -> lrelu -> conv(k,1) -> +x) run as ONE fused launch where the kernel
        # covers the shape (C in {8,16,32,64,128,256} per `fuse_channels`, split-bf16 / two-part fp16 arithmetic): the intermediate tensor stays in
        # LDS, 5 HBM tensor passes -> 2.  Bitwise equal to the unfused pair.
        # Measured at the benchmark's stage shapes (scripts/resblock_ab.py, fused / unfused time): C=32 0.59-0.83,
        # C=64 0.66-0.92, C=128 0.88 (k=3), 1.00 (k=7), 1.07 (k=11) -> the 128-channel stage fuses its k=3 blocks only.
        self.fuse_resblocks = os.environ.get("TTSAMD_FUSE_RESBLOCKS", "1") != "0"
        self.fuse_channels = tuple(int(c) for c in os.environ.get("TTSAMD_FUSE_CHANNELS", "8,16,32,64,128,256").split(",") if c)
        # channel count -> largest kernel size fused (absent = all).  128 channels: k = 3 only on six products (round 2: k = 7 1.00,
        # k = 11 1.07 of the unfused pair); on three products the fused k = 7 pair is 0.90-0.91 of the two launches (its LDS image is
        # 2/3 the size), k = 11 1.02 (scripts/h2_variants_ab.py) — `None` = pick by the conv precision at call time.
        # 256 channels (three products only, an 8-wave block per CU): launch by launch k = 3 is 0.85-0.92 of the two launches at
        # B = 32 and 0.70-0.73 for a single utterance, k = 7 1.06-1.08 / 0.90, k = 11 1.20 / 1.00 (scripts/r6_pair256_ab.py); in the
        # whole B = 32 step (same box, TTSAMD_FUSE_LIMITS): none 49.04, k <= 3 48.67, k <= 7 48.33, all 49.14 ms -> k <= 7
        # (profiles/r06_pair256_ab.txt; the step gains more than the launches: one tensor pass less per pair under the power cap)
        self.fuse_max_kernel = None
        # small grids (a single sentence): the three MRF branches of a stage as ONE launch per ResBlock iteration instead of nine
        # launches on three branch streams (forward())
        self.group_branches = os.environ.get("TTSAMD_GROUP_BRANCHES", "1") != "0"
        self._side_streams = collections.OrderedDict()     # current stream handle -> its MRF branch streams (LRU first)
        self._retired = []          # evicted sets: parked, destroyed only by release_streams()
        self._torch_streams = set()
        # A single utterance through the vocoder is ~110 launches of a few microseconds each on up to four streams: issued
        # one by one the host is the bottleneck.  `inference` on one item of up to `graph_max_frames` frames replays as ONE
        # hipGraph per 32-frame length bucket (the item runs ragged-exact inside the padded tensor: same samples).
        self.use_graphs = True
        self.graph_max_frames = 2048
        # A single item through `inference` runs behind the model-level C handle (ttsamd_hifigan_*, csrc/hifigan_model.hip): weights
        # folded HERE handed over once, then per request one staging copy, ONE C call (the handle replays its hipGraph of the
        # 32-frame bucket, ragged-exact) and one clone — measured 320 vs 361 us per sentence against this class's own graph replay
        # (profiles/r05_native_vocoder_ab.txt).  Speaker-conditioned generators and batches stay on the Python host below.
        # TTSAMD_NATIVE_MODELS=0 / use_native = False: Python host everywhere.
        self.use_native = os.environ.get("TTSAMD_NATIVE_MODELS", "1") != "0"
        self._native = {}                      # stream handle -> (NativeHifigan, {t_pad: (mel, lengths, wav) static buffers})
        self._graph = graphs.GraphCache(self._inference_ragged, max_entries=12)
        self.weights_version = 0    # bumped by every re-pack: dependants (SentencePipeline) key their graphs on it

    def _fuse_limit(self, ch, cols=None):
        """Largest kernel size whose ResBlock pairs run fused at `ch` channels (see fuse_max_kernel); `cols` = T x batch: the
        256-channel pair is a large-grid tile (an 8-wave block per CU) and stays off below 4096 columns."""
        if ch == 256 and cols is not None and cols < 4096:
            return 0
        table = self.fuse_max_kernel if self.fuse_max_kernel is not None else ({128: 7, 256: 7} if ops.conv_precision() == "h2" else {128: 3, 256: 0})
        env = os.environ.get("TTSAMD_FUSE_LIMITS")         # "128:7,256:7": A/B override, read by the C handle too
        if env and self.fuse_max_kernel is None:
            table = {**table, **{int(c): int(k) for c, k in (item.split(":") for item in env.split(",") if item)}}
            if ops.conv_precision() != "h2":
                table[256] = 0
        return table.get(ch, 99)

    def hop_length(self):
        return _cumprod(self.upsample_factors)[-1]

    def context_frames(self):
        """Input frames either side that can influence one output sample (see tts_amd/xtts_stream.py)."""
        from .xtts_stream import context_frames
        return context_frames(self.upsample_factors, self.resblock_type, self.resblock_kernel_sizes,
                              self.resblock_dilation_sizes)

    # ---- torch.nn.Module-like surface used by Synthesizer (synthesizer.py:222-225,379) -----------
    def parameters(self):
        if self._packed is None:
            return iter(())
        return iter([self._packed["conv_pre"].w])

    def eval(self):
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def to(self, device):
        self.device = torch.device(device)
        if self._sd is not None:
            self._pack()
        return self

    def remove_weight_norm(self):  # folding happens at pack time; kept for API parity (:284-291)
        pass

    def state_dict(self):
        return dict(self._sd or {})

    def load_state_dict(self, sd, strict=True, prefix=""):
        self._sd = {k[len(prefix):]: v.detach().cpu() for k, v in sd.items() if k.startswith(prefix)}
        if self.device.type == "cuda":
            self._pack()

    def load_checkpoint(self, config, checkpoint_path, eval=False, cache=False):  # noqa: A002
        """hifigan_generator.py:293-301: `{"model": state_dict}` checkpoints."""
        state = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        self.load_state_dict(state["model"])

    # ---- weight preparation (load-time glue) ------------------------------------------------------
    def _pack(self):
        if self.device.type != "cuda":
            raise _lib.TtsAmdError("tts_amd.HifiganGenerator runs only on a GPU (no CPU fallback)")
        sd, dev = self._sd, self.device
        P = {}
        P["conv_pre"] = PackedConv(ops.fold_weight_norm(sd, "conv_pre"), sd.get("conv_pre.bias"), dev)
        if self.cond_channels > 0 and "cond_layer.weight" in sd:
            P["cond_layer"] = PackedConv(sd["cond_layer.weight"], sd.get("cond_layer.bias"), dev)
        for i, u in enumerate(self.upsample_factors):
            w, b = ops.convt_polyphase_weight(ops.fold_weight_norm(sd, "ups.%d" % i), sd.get("ups.%d.bias" % i), u)
            P["ups.%d" % i] = PackedConv(w, b, dev, pad_left=w.shape[2] - 1)
            if self.cond_in_each_up_layer and ("conds.%d.weight" % i) in sd:
                # conds[i] is a 1x1 conv of g: its output (one offset per (b, channel)) rides in the transposed conv's
                # epilogue; rows are repeated per polyphase row (packed row = channel*u + phase)
                P["conds.%d" % i] = PackedConv(sd["conds.%d.weight" % i].repeat_interleave(u, 0),
                                               sd["conds.%d.bias" % i].repeat_interleave(u, 0), dev)
        for i in range(self.num_upsamples):
            for j, (k, dil) in enumerate(zip(self.resblock_kernel_sizes, self.resblock_dilation_sizes)):
                rp = "resblocks.%d." % (i * self.num_kernels + j)
                for m, d in enumerate(dil):
                    if self.resblock_type == "1":
                        P[rp + "convs1.%d" % m] = PackedConv(ops.fold_weight_norm(sd, rp + "convs1.%d" % m),
                                                             sd.get(rp + "convs1.%d.bias" % m), dev, dilation=d)
                        P[rp + "convs2.%d" % m] = PackedConv(ops.fold_weight_norm(sd, rp + "convs2.%d" % m),
                                                             sd.get(rp + "convs2.%d.bias" % m), dev, dilation=1)
                    else:
                        P[rp + "convs.%d" % m] = PackedConv(ops.fold_weight_norm(sd, rp + "convs.%d" % m),
                                                            sd.get(rp + "convs.%d.bias" % m), dev, dilation=d)
        P["conv_post"] = PackedConv(ops.fold_weight_norm(sd, "conv_post"), sd.get("conv_post.bias"), dev)
        self._graph.clear()          # captured graphs hold raw pointers to the previous weight tensors
        self.weights_version += 1
        self._drop_native()
        self._packed = P

    def _drop_native(self):
        for nat, _ in self._native.values():
            nat.close()
        self._native = {}

    def _native_single(self, c):
        """One item [1, C, T] behind the vocoder handle: staged into the 32-frame bucket's static buffers, replayed, cloned."""
        key = torch.cuda.current_stream().cuda_stream
        ent = self._native.get(key)
        if ent is None:
            sd = {}
            for k, v in self._sd.items():
                if k.endswith(".parametrizations.weight.original0") or k.endswith(".weight_g"):
                    name = k[: -len(".parametrizations.weight.original0")] if k.endswith("original0") else k[: -len(".weight_g")]
                    sd[name + ".weight"] = ops.fold_weight_norm(self._sd, name)
                elif not (k.endswith(".parametrizations.weight.original1") or k.endswith(".weight_v")):
                    sd[k] = v
            if len(self._native) >= 4:
                self._native.pop(next(iter(self._native)))[0].close()
            ent = self._native[key] = (NativeHifigan(self, sd), {})
        nat, bufs = ent
        T = c.shape[2]
        t_pad = -(-T // 32) * 32
        b = bufs.get(t_pad)
        if b is None:
            if len(bufs) >= 12:
                bufs.pop(next(iter(bufs)))
            b = bufs[t_pad] = (torch.zeros((1, c.shape[1], t_pad), dtype=torch.float32, device=c.device),
                               torch.empty((1,), dtype=torch.int64, device=c.device),
                               torch.empty((1, self.out_channels, nat.output_samples(t_pad)), dtype=torch.float32, device=c.device))
        mel, lens, wav = b
        ops.copy_into([mel[:, :, :T]], [c])            # (frames beyond T keep an earlier request's values: masked out by `lengths`)
        lens.fill_(T)
        nat.set_concurrent_branches((parallel.active_lanes() <= 1) if self.concurrent_branches == "auto" else bool(self.concurrent_branches))
        nat.forward(mel, lens, use_graph=True, out=wav)
        hop = wav.shape[-1] // (t_pad + 2 * self.inference_padding)
        return wav[:, :, : (T + 2 * self.inference_padding) * hop].clone()      # the handle's buffer is static: hand out a copy

    def _group_stage(self, i, ch, B, T, P):
        """Can stage i's MRF run as grouped launches (ops.resblock_group)?  All branches ResBlock1 with the same dilation list,
        fusable pairs of distinct kernel sizes from {3, 7, 11}, a small-grid shape, 2-3 branches."""
        nk = self.num_kernels
        dils = self.resblock_dilation_sizes[:nk]
        if not (self.fuse_resblocks and ch in self.fuse_channels and 2 <= nk <= 3 and T % 4 == 0 and all(d == dils[0] for d in dils)):
            return False
        for m in range(len(dils[0])):
            pairs = [(P["resblocks.%d.convs1.%d" % (i * nk + j, m)], P["resblocks.%d.convs2.%d" % (i * nk + j, m)]) for j in range(nk)]
            if any(pc1.kernel > self._fuse_limit(ch, B * T) for pc1, _ in pairs) or not ops.resblock_group_supported(pairs, B, ch, T):
                return False
        return True

    def weight_bytes(self):
        return sum(p.nbytes() for p in self._packed.values())

    def _streams(self, n):
        """MRF branch streams of the CURRENT stream: every request lane (parallel.Lanes) and every capture stream gets its
        own set, so that two requests in flight never queue work on a common stream (round 2 shared one set between the
        lanes: one lane's branch launches then sat behind the other lane's event waits).  Sets are kept least-recently-used
        first; beyond 16 owner streams the oldest set whose owner stream is GONE is retired — a set whose owner is alive
        (another thread may be in the middle of forward() on it) is never evicted, and a retired set's streams are only
        parked (`_retired`), not destroyed, until `release_streams()`: a view handed out earlier stays valid."""
        key = torch.cuda.current_stream().cuda_stream
        pool = self._side_streams.get(key)
        if pool is None:
            if len(self._side_streams) >= 16:
                for k in list(self._side_streams):
                    if k != 0 and _lib.stream_owner(k) is None and k not in self._torch_streams:
                        self._retired.append(self._side_streams.pop(k))
                        if len(self._side_streams) < 16:
                            break
            pool = self._side_streams[key] = []
            if _lib.stream_owner(key) is None and key != 0:
                self._torch_streams.add(key)          # a torch-pool stream: never destroyed, its set is never retired
        else:
            self._side_streams.move_to_end(key)
        while len(pool) < n:
            pool.append(_lib.OwnedStream(self.device))
        return [o.stream for o in pool[:n]]

    def release_streams(self, handle=None):
        """Tear-down hook (a serving loop that rebuilds its lanes): after a device-wide wait, destroy the parked branch-stream
        sets, the set keyed by `handle` (or every set when None), and the graphs captured for that stream."""
        torch.cuda.synchronize()
        del self._retired[:]
        if handle is None:
            self._side_streams.clear()
            self._graph.clear()
        else:
            self._side_streams.pop(handle, None)
            self._graph.purge_stream(handle)

    # ---- forward (hifigan_generator.py:236-265) ----------------------------------------------------
    @torch.no_grad()
    def forward(self, x, g=None, in_mask=None, lengths=None, _masks=None):
        """`in_mask` [B,T] (optional) multiplies x inside conv_pre's load: VITS feeds `z * y_mask` (vits.py:1161).
        `lengths` [B] (optional, frames): ragged-exact batching — every conv of every stage reads item b as if its
        tensor ended at lengths[b] (positions beyond are zero, exactly the zero padding a stand-alone run of that item
        sees), so the first lengths[b]*hop samples of row b equal a B=1 run on that item alone — bit for bit when every
        conv launch of the two runs takes the same tile family, to fp32 reassociation (~1e-6 relative) otherwise: the
        launcher picks the small-grid kernels (K loop split over wave groups, a different fp32 summation order) from the
        launch's total block count, which grows with the batch (conv_kernel_x3.h: conv1d_x3_launch_tiles)."""
        if self._packed is None:
            raise _lib.TtsAmdError("HifiganGenerator: no weights loaded / not moved to the GPU")
        _lib.require_gpu(x, "x")
        P = self._packed
        x = x.contiguous().float()
        B, _, T = x.shape
        dev = x.device
        new = lambda c, t: torch.empty((B, c, t), dtype=torch.float32, device=dev)  # noqa: E731
        ch = self.upsample_initial_channel
        o = new(ch, T)
        sm = [None] * (self.num_upsamples + 1)        # per-stage length masks [B, T_stage]
        if (lengths is not None or _masks is not None) and not self.exact_hop:
            raise _lib.TtsAmdError("ragged batching needs upsample kernels with k - stride even (output = frames * hop exactly)")
        if _masks is not None:
            sm = list(_masks)
            in_mask = sm[0] if in_mask is None else in_mask * sm[0]
        elif lengths is not None:
            # every stage's mask in ONE launch (a sequence_mask + a lengths*scale per stage before: 9 tiny launches at the head
            # of every ragged call)
            scales = [1] + _cumprod(self.upsample_factors)
            sm, _ = ops.stage_masks(lengths.to(dev), scales, [T * sc for sc in scales])
            in_mask = sm[0] if in_mask is None else in_mask * sm[0]
        if g is not None and "cond_layer" in P:
            # o = conv_pre(x) + cond_layer(g): g is [B, C, 1]; its 1x1 conv is a per-(b, channel) offset that
            # rides in conv_pre's epilogue (hifigan_generator.py:250-251)
            gc = new(ch, 1)
            ops.conv1d(P["cond_layer"], g.contiguous().float(), gc)
            ops.conv1d(P["conv_pre"], x, o, in_mask=in_mask, row_bias=gc.reshape(B, ch))
        else:
            ops.conv1d(P["conv_pre"], x, o, in_mask=in_mask)
        nk = self.num_kernels
        concurrent = (parallel.active_lanes() <= 1) if self.concurrent_branches == "auto" else bool(self.concurrent_branches)
        for i, u in enumerate(self.upsample_factors):
            ch //= 2
            k_up = self.upsample_kernel_sizes[i]
            pad_up = (k_up - u) // 2                                  # hifigan_generator.py:216
            T_up = (T - 1) * u - 2 * pad_up + k_up                    # = T * u when k - u is even
            up = new(ch, T_up)
            rb = ops.speaker_cond(P["conds.%d" % i], g) if (g is not None and ("conds.%d" % i) in P) else None
            ops.conv1d(P["ups.%d" % i], o, up, in_act=ACT_LRELU, in_slope=LRELU_SLOPE, mode=CONV_SHUFFLE,
                       shuffle_u=u, shuffle_pad=pad_up, in_mask=sm[i], row_bias=rb, t_out=T + P["ups.%d" % i].kernel - 1)
            msk = sm[i + 1]
            T = T_up
            o_next = new(ch, T)
            zsum = new(ch, T) if nk > 1 else None
            # The MRF's resblocks are independent until their last conv (which chains the accumulate r1 + r2 + r3 in
            # the reference's order): each branch runs on its own HIP stream so that one branch's launch tail / ramp
            # overlaps another branch's compute; events order only the accumulating convs.
            if self.group_branches and self.resblock_type == "1" and self._group_stage(i, ch, B, T, P):
                # a single sentence (small grids): the stage's branches as ONE launch per iteration on this stream, their
                # outputs averaged by one more (ttsamd_resblock_group, ttsamd_sum_div) — no branch streams, no joins
                dil = self.resblock_dilation_sizes[0]
                cur = [up] * nk
                bufs = [(new(ch, T), new(ch, T)) for _ in range(nk)]
                for m in range(len(dil)):
                    pairs = [(P["resblocks.%d.convs1.%d" % (i * nk + j, m)], P["resblocks.%d.convs2.%d" % (i * nk + j, m)]) for j in range(nk)]
                    dsts = [bufs[j][m & 1] for j in range(nk)]
                    ops.resblock_group(pairs, cur, dsts, slope=LRELU_SLOPE, mask=msk)
                    cur = dsts
                if T % 4 == 0:
                    ops.sum_div(cur, o_next, float(nk))
                    o = o_next
                    continue
                raise _lib.TtsAmdError("grouped MRF stage with T % 4 != 0")     # _group_stage excludes it
            main = torch.cuda.current_stream()
            side = self._streams(nk) if concurrent and nk > 1 else None
            ev_up = torch.cuda.Event() if side else None
            if side:
                ev_up.record(main)
            prev_done = None
            keep = []   # branch buffers stay referenced until the branches are joined (allocator reuse is per stream)
            for j in range(nk):
                rp = "resblocks.%d." % (i * nk + j)
                dil = self.resblock_dilation_sizes[j]
                xa, xb = new(ch, T), new(ch, T)
                tmp = None               # conv1 -> conv2 intermediate of an UNFUSED ResBlock1 iteration (allocated on first use)
                keep += [xa, xb]
                st = side[j] if side else main
                if side:
                    st.wait_event(ev_up)
                with torch.cuda.stream(st):
                    cur = up
                    for m in range(len(dil)):
                        last = m == len(dil) - 1
                        if last:  # fuse the MRF accumulate / average into the block's last conv
                            dst = o_next if j == nk - 1 else zsum
                            accum = zsum if j > 0 else None
                            div = float(nk) if j == nk - 1 else 0.0
                        else:
                            dst, accum, div = (xa if cur is not xa else xb), None, 0.0
                        if self.resblock_type == "1":
                            pc1, pc2 = P[rp + "convs1.%d" % m], P[rp + "convs2.%d" % m]
                            if (self.fuse_resblocks and ch in self.fuse_channels and pc1.kernel <= self._fuse_limit(ch, B * T)
                                    and ops.resblock_pair_supported(pc1, pc2)):
                                if last and side and prev_done is not None:
                                    st.wait_event(prev_done)      # zsum holds the previous branches' sum
                                ops.resblock_pair(pc1, pc2, cur, dst, slope=LRELU_SLOPE, mask=msk, accum=accum, out_div=div)
                                cur = dst
                                continue
                            if tmp is None:
                                tmp = new(ch, T)
                                keep.append(tmp)
                            ops.conv1d(pc1, cur, tmp, in_act=ACT_LRELU, in_slope=LRELU_SLOPE, in_mask=msk)
                            if last and side and prev_done is not None:
                                st.wait_event(prev_done)      # zsum holds the previous branches' sum
                            ops.conv1d(pc2, tmp, dst, in_act=ACT_LRELU, in_slope=LRELU_SLOPE,
                                       res=cur, accum=accum, out_div=div, in_mask=msk)
                        else:
                            if last and side and prev_done is not None:
                                st.wait_event(prev_done)
                            ops.conv1d(P[rp + "convs.%d" % m], cur, dst, in_act=ACT_LRELU, in_slope=LRELU_SLOPE,
                                       res=cur, accum=accum, out_div=div, in_mask=msk)
                        cur = dst
                    if side:
                        prev_done = torch.cuda.Event()
                        prev_done.record(st)
            if side:
                main.wait_event(prev_done)       # the last branch's final conv completes the chain
            del keep
            o = o_next
        wav = new(self.out_channels, T)
        # final F.leaky_relu(o) uses the DEFAULT slope 0.01 (hifigan_generator.py:262)
        ops.conv1d(P["conv_post"], o, wav, in_act=ACT_LRELU, in_slope=0.01, out_act=ACT_TANH, in_mask=sm[-1])
        return wav

    __call__ = forward

    @torch.no_grad()
    def inference_slabbed(self, c, out=None, max_live_bytes=48 << 30):
        """`inference` over a batch too large to hold layer-by-layer (BASELINE config 3: [256, 80, 8192] mels would need
        6 live tensors of 69 GB each): the batch is cut into slabs whose live activations fit `max_live_bytes`; items are
        independent, so slabbing changes nothing but — when a slab is small enough for the launcher to pick the small-grid
        tile family where the full batch would not — the fp32 summation order inside a conv (~1e-6 relative).  c may live on the host or the device; `out`
        ([B,1,(T+2p)*hop], host or device) receives the waveforms (allocated on c's device if None)."""
        B, C, T = c.shape
        hop = 1
        for u in self.upsample_factors:
            hop *= u
        t_out = (T + 2 * self.inference_padding) * hop
        # peak live set per item: ~6 tensors of [C0/2, T*u0] floats at the widest stage (up, tmp, xa, xb, zsum, o_next)
        widest = max((self.upsample_initial_channel >> (i + 1)) * hop_i for i, hop_i in
                     enumerate(_cumprod(self.upsample_factors)))
        per_item = 6 * 4 * widest * (T + 2 * self.inference_padding)
        slab = max(1, min(B, int(max_live_bytes // max(per_item, 1))))
        if out is None:
            out = torch.empty((B, self.out_channels, t_out), dtype=torch.float32, device=c.device)
        for lo in range(0, B, slab):
            hi = min(B, lo + slab)
            w = self._inference_ragged(c[lo:hi].to(self.device, non_blocking=True).contiguous().float())   # throughput path: no graphs
            out[lo:hi].copy_(w, non_blocking=True)
        return out

    @torch.no_grad()
    def inference(self, c, lengths=None):
        """hifigan_generator.py:267-282: replicate-pad `inference_padding` frames each side, no crop.
        `lengths` [B] (frames, optional) = ragged-exact batching (see forward): row b's first
        (lengths[b] + 2*pad)*hop samples equal `inference(c[b:b+1, :, :lengths[b]])`."""
        c = c.to(self.device).contiguous().float()
        if (self.use_graphs and self.exact_hop and lengths is None and c.shape[0] == 1 and 0 < c.shape[2] <= self.graph_max_frames):
            if self.use_native and not (self.cond_channels > 0 or self.cond_in_each_up_layer) and self._sd is not None:
                return self._native_single(c)
            T = c.shape[2]
            t_pad = -(-T // 32) * 32
            cp = torch.zeros((1, c.shape[1], t_pad), dtype=torch.float32, device=c.device)
            cp[:, :, :T] = c
            wav = self._graph(cp, torch.full((1,), T, dtype=torch.int64, device=c.device))
            hop = wav.shape[-1] // (t_pad + 2 * self.inference_padding)
            return wav[:, :, : (T + 2 * self.inference_padding) * hop].clone()     # the graph's buffer is static: hand out a copy
        return self._inference_ragged(c, lengths)

    def _inference_ragged(self, c, lengths=None, quantum=1):
        """`quantum` > 1: item b's frame count is lengths[b] // quantum * quantum (a Glow-TTS mel whose squeeze dropped the
        frames that did not fill a group: the count comes straight from the model's y_lengths, on the device)."""
        p = self.inference_padding
        if lengths is None:
            if p > 0:
                B, C, T = c.shape
                cp = torch.empty((B, C, T + 2 * p), dtype=torch.float32, device=c.device)
                ops.replicate_pad(c, cp, p)
                c = cp
            return self.forward(c)
        lengths = torch.as_tensor(lengths).to(self.device, torch.int64)     # masks and pad read them on the device
        B, C, T = c.shape
        scales = [1] + _cumprod(self.upsample_factors)
        masks, len_eff = ops.stage_masks(lengths, scales, [(T + 2 * p) * sc for sc in scales], quantum=quantum, add=2 * p)
        if p > 0:
            cp = torch.empty((B, C, T + 2 * p), dtype=torch.float32, device=c.device)
            ops.replicate_pad(c, cp, p, len_eff, len_bias=-2 * p)
            c = cp
        return self.forward(c, _masks=masks)
