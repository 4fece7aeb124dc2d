"""`Synthesizer`-compatible inference driver (TTS/utils/synthesizer.py:25-505) over the HIP models.

Same constructor arguments and `tts()` / `save_wav()` / `split_into_sentences()` contract for the model
families this build covers (`vits`, `glow_tts` + `hifigan` GAN vocoder); configs are the reference's JSON files
read as plain dicts (coqpit is not needed).  Differences, all on the fast side:
  * sentences of one request are synthesised as ONE padded batch (the reference loops B=1, synthesizer.py:384);
  * the Glow-TTS -> vocoder mel seam (denormalize / normalize, synthesizer.py:412-429) runs on the device;
  * waveforms come back as numpy arrays and are concatenated once (`tts()` still returns a flat list of floats
    like the reference; `tts_batch()` is the array API).
"""
import json
import re
import time

import numpy as np
import torch

from . import _lib
from .audio import AudioProcessor, mel_renorm_device
from .gan import GAN
from .glow_tts import GlowTTS
from .managers import LanguageManager, SpeakerManager
from .text import TTSTokenizer
from .vits import Vits, _get

_MODELS = {"vits": Vits, "glow_tts": GlowTTS}
# pysbd's English tables (restated).  Titles that precede a name (PREPOSITIVE_ABBREVIATIONS): a period after them never
# ends a sentence.  Abbreviations that precede a NUMBER ("fig. 2", "no. 5") stay whole only before a digit.  Only the
# MULTI-PERIOD abbreviations pysbd restores around its sentence starters (U.S, U.K, E.U, U.S.A, I.V, U.N) end a sentence only
# when one of those starters follows ("U.S. Army" stays whole, "... in the U.S. The next day ..." splits); "e.g." / "i.e."
# never end one.  Every other word — ordinary words that happen to be abbreviations too ("no", "sat", "mar", "etc", "inc",
# "p.m") — ends a sentence before any token that does not start with a lowercase letter.
_NUMBER_ABBREVIATIONS = frozenset("art ext no nos p pp fig figs vol vols ch sec eq para".split())
_STARTER_ABBREVIATIONS = frozenset("u.s u.k e.u u.s.a u.n i.v".split())
_NEVER_FINAL_ABBREVIATIONS = frozenset("e.g i.e".split())
_SENTENCE_STARTERS = frozenset("A Being Did For He How However I In It Millions More She That The There They We What When "
                               "Where Who Why".split())
_PREPOSITIVE_ABBREVIATIONS = frozenset("adm attys brig capt cmdr col cpl det dr gen gov ing lt maj mr mrs ms mt messrs mssrs prof ph rep reps rev sen sens sgt st supt v vs".split())


def load_config(path):
    """config/__init__.py:68-100 without coqpit: JSON (comments tolerated, :14-21) -> dict."""
    txt = open(path, "r", encoding="utf-8").read()
    try:
        return json.loads(txt)
    except json.JSONDecodeError:
        txt = re.sub(r"\\\n", "", txt)
        txt = re.sub(r"//.*\n", "\n", txt)
        return json.loads(txt)


def setup_tts_model(config):
    """tts/models/__init__.py:6-14: class found by `config["model"]`."""
    name = str(_get(config, "model", "")).lower()
    if name not in _MODELS:
        raise _lib.TtsAmdError("no HIP implementation for tts model %r (have: %s)" % (name, sorted(_MODELS)))
    cfg = dict(config) if isinstance(config, dict) else config
    ap = AudioProcessor.init_from_config(cfg)
    tok, _ = TTSTokenizer.init_from_config(cfg)
    if isinstance(cfg, dict):
        cfg = dict(cfg, _ap=ap, _tokenizer=tok, _speaker_manager=SpeakerManager.init_from_config(cfg),
                   _language_manager=LanguageManager.init_from_config(cfg))
        if name == "glow_tts" and cfg.get("num_chars") is None:
            cfg["num_chars"] = tok.characters.num_chars
    return _MODELS[name].init_from_config(cfg)


class SentencePipeline:
    """Glow-TTS -> mel seam -> HiFiGAN vocoder for one request (a sentence, or a ragged-exact batch of sentences) as TWO graph
    replays around the request's one host wait — the reference runs `tts_model.inference`, two numpy normalisations and
    `vocoder_model.inference` back to back per sentence (synthesizer.py:384-433):
        graph 1   encoder + duration predictor                                   (GlowTTS.request_front)
        eager     durations kernel; the host polls the output extent from a pinned mirror
        graph 2   prior expansion -> 12 decoder flow blocks -> denormalize/normalize seam -> replicate pad + stage masks ->
                  every vocoder conv, at the frame count padded to a multiple of 32 (the sentence runs ragged-exact inside)
        eager     ONE copy of the valid samples out of the graph's fixed output buffer.
    Nothing crosses to the host between the acoustic model and the vocoder; the mel never leaves channels-first layout."""

    def __init__(self, tts_model, vocoder_g, tts_ap, vocoder_ap):
        from . import graphs

        self.tts, self.voc, self.ap_t, self.ap_v = tts_model, vocoder_g, tts_ap, vocoder_ap
        self._graph = graphs.GraphCache(self._tail_eager, max_entries=12)
        # the captured tail holds raw pointers to BOTH models' weight tensors and reads the acoustic model's per-stream scratch in
        # place: it is keyed on the models' weight versions (a re-pack — .cuda(), load_checkpoint, a hot swap — drops it) and
        # registered with the scratch so that evicting a scratch set drops the graphs over it
        self._weights = None
        if hasattr(tts_model, "_scratch"):
            tts_model._scratch.dependents.append(self._graph)
        self._cfg = None
        self.max_frames = 4096
        self.launches = None       # kernel launches of the last captured tail (reported by bench.py)
        # request_front's context of a request that did not fit the fused path (handed to inference): per host thread — two threads
        # serving through one Synthesizer must not pick up each other's front end
        import threading

        self._tls = threading.local()

    @property
    def last_ctx(self):
        return getattr(self._tls, "ctx", None)

    @last_ctx.setter
    def last_ctx(self, ctx):
        self._tls.ctx = ctx

    def supported(self):
        return isinstance(self.tts, GlowTTS) and self.tts.use_graphs and self.voc is not None

    def clear(self):
        self._graph.clear()

    def purge_stream(self, handle):
        """Lanes.close: drop the graphs captured for a request lane that is being torn down."""
        self._graph.purge_stream(handle)

    def _check_weights(self):
        v = (getattr(self.tts, "weights_version", 0), getattr(self.voc, "weights_version", 0))
        if v != self._weights:
            if self._weights is not None:
                self._graph.clear()
            self._weights = v
        return v

    def _tail_eager(self, o_mean, o_logs, cum, x_mask, y_lengths, noise, g):
        from . import ops

        t_pad, noise_scale = self._cfg
        pri = ops.expand_prior(o_mean, o_logs if o_logs.numel() else None, noise if noise.numel() else None, cum, x_mask,
                               y_lengths, t_pad, noise_scale, mask_out=True, want_stats=False, noise_packed=True)
        mel = self.tts.decoder(pri["z_p"], pri["y_mask"], g=g if g.numel() else None)            # [B, C, t_pad]
        voc_in = mel_renorm_device(mel, self.ap_t, self.ap_v)                                    # synthesizer.py:412-416
        # the vocoder reads every item as (y_lengths // num_squeeze) * num_squeeze frames long (+ its replicate padding)
        return self.voc._inference_ragged(voc_in, y_lengths, quantum=self.tts.num_squeeze)

    @torch.no_grad()
    def __call__(self, x, aux_input=None, eager=False):
        """-> (wav float32 [B, 1, max valid samples] on the device, [valid samples per item]) or None when the request does
        not fit the fused path (the caller then runs the models one after the other).  eager=True issues the same launch
        sequence one by one instead of replaying graphs (measurement / debugging)."""
        from . import ops

        tts = self.tts
        wv = self._check_weights()
        ctx = tts.request_front(x, dict(aux_input or {}, no_graph=True) if eager else aux_input)
        B, t_dec = ctx["B"], ctx["t_dec"]
        t_pad = -(-t_dec // 32) * 32
        if not ((ctx["graphing"] or eager) and (B == 1 or ctx["ragged"]) and B * t_pad <= self.max_frames):
            # too large for the fused path: the caller continues from the front end that has already run (no second encoder /
            # duration-predictor pass, no second host wait)
            self.last_ctx = ctx
            return None
        t_pad, inputs, stable = tts.tail_inputs(ctx, aux_input)
        self._cfg = (t_pad, float(tts.inference_noise_scale))
        self._graph.enabled = not eager
        wav = self._graph(*inputs, key=self._cfg + wv, stable=stable)
        nsq, pad = tts.num_squeeze, self.voc.inference_padding
        hop = wav.shape[-1] // (t_pad + 2 * pad)
        lens = [((n // nsq) * nsq + 2 * pad) * hop for n in ctx["y_lengths_host"]]
        return ops.clone_views([wav[:, :, : max(lens)]])[0], lens


class Synthesizer:
    def __init__(self, tts_checkpoint="", tts_config_path="", tts_speakers_file="", tts_languages_file="",
                 vocoder_checkpoint="", vocoder_config="", encoder_checkpoint="", encoder_config="", vc_checkpoint="",
                 vc_config="", model_dir="", voice_dir=None, use_cuda=True):
        if not use_cuda or not torch.cuda.is_available():
            raise _lib.TtsAmdError("tts_amd.Synthesizer needs a GPU (use_cuda=True): there is no CPU path")
        self.use_cuda = True
        self.tts_speakers_file, self.tts_languages_file = tts_speakers_file, tts_languages_file
        self.voice_dir = voice_dir
        self.tts_config = load_config(tts_config_path) if isinstance(tts_config_path, str) else tts_config_path
        if isinstance(self.tts_config, dict) and (tts_speakers_file or tts_languages_file):
            # what ModelManager._update_paths does to a downloaded config (utils/manage.py:455-500): point the config's
            # speaker / language files at the ones given here, so the managers built from the config read them
            self.tts_config = dict(self.tts_config)
            margs = dict(self.tts_config.get("model_args") or {})
            look = lambda k: margs.get(k, self.tts_config.get(k))  # noqa: E731
            if tts_speakers_file:
                margs["d_vector_file" if look("use_d_vector_file") else "speakers_file"] = tts_speakers_file
            if tts_languages_file:
                margs["language_ids_file"] = tts_languages_file
            self.tts_config["model_args"] = margs
        self.tts_model = setup_tts_model(self.tts_config)
        if tts_checkpoint:
            self.tts_model.load_checkpoint(self.tts_config, tts_checkpoint, eval=True)
        self.tts_model.cuda()
        self.output_sample_rate = _get(_get(self.tts_config, "audio", {}), "sample_rate", 22050)
        self.vocoder_model = self.vocoder_ap = self.vocoder_config = None
        if vocoder_checkpoint or vocoder_config:
            self.vocoder_config = load_config(vocoder_config) if isinstance(vocoder_config, str) else vocoder_config
            self.vocoder_ap = AudioProcessor.init_from_config(self.vocoder_config)
            self.vocoder_model = GAN.init_from_config(self.vocoder_config)
            if vocoder_checkpoint:
                self.vocoder_model.load_checkpoint(self.vocoder_config, vocoder_checkpoint, eval=True)
            self.vocoder_model.cuda()
            self.output_sample_rate = _get(_get(self.vocoder_config, "audio", {}), "sample_rate", self.output_sample_rate)
        self.device = next(self.tts_model.parameters()).device if self.tts_model._sd is not None else torch.device("cuda")
        self.pipeline = None
        if self.vocoder_model is not None:
            self.pipeline = SentencePipeline(self.tts_model, self.vocoder_model.model_g, self.tts_model.ap, self.vocoder_ap)

    @staticmethod
    def split_into_sentences(text):
        """synthesizer.py:227-236 segments with `pysbd.Segmenter(language="en", clean=True)`; pysbd is not in this image,
        so this is a rule-based restatement of the English behaviour the reference's own test pins
        (tests/inference_tests/test_synthesizer.py:29-79 — all 17 golden strings are checked in tests/test_host_cpu.py):
          * a boundary is sentence-final punctuation (+ closing quotes / brackets) followed by whitespace and a token that
            does not start with a lowercase letter;
          * titles that precede a name (dr., mr., mrs., ...) never end a sentence; other abbreviations (co., jr., U.K.)
            do when a capitalised word follows — pysbd's prepositive / other abbreviation split;
          * runs of list markers `1.) 2.)`, `1) 2)`, `1. 2.`, `a. b. c.` start a new segment each.
          * abbreviations before a number ("fig. 2", "no. 5") and the multi-period abbreviations U.S / U.K / E.U / U.S.A / U.N /
            I.V before a word that is not one of pysbd's sentence starters ("The U.S. Army is big.") do not end a sentence;
            ordinary words that double as abbreviations ("no.", "sat.", "etc.", "p.m.") do, before any capitalised token.
        Known differences from pysbd: no per-language rule sets (`_get_segmenter(lang)`), no ellipsis / parenthetical /
        exclamation-word ("Yahoo!") tables beyond the lowercase-follows rule."""
        text = re.sub(r"\s+", " ", text.strip())
        if not text:
            return []
        # ---- list markers: consecutive 1,2,3.. or a,b,c.. at token starts -------------------------------------------
        cuts = []
        for rx, seq in ((r"(?:(?<=\s)|^)(\d{1,2})(?:\.\)|\)|\.)(?=\s)", lambda k: str(k + 1)),
                        (r"(?:(?<=\s)|^)([a-z])(?:\.|\))(?=\s)", lambda k: chr(ord("a") + k))):
            ms, k = [], 0
            for m in re.finditer(rx, text):
                if m.group(1) == seq(k):
                    ms.append(m)
                    k += 1
            if len(ms) >= 2:
                cuts = [(m.start(), m.end()) for m in ms]
                break
        segments = []
        if cuts:
            if cuts[0][0] > 0:
                segments.append((text[: cuts[0][0]], 0))
            for i, (a, b) in enumerate(cuts):
                end = cuts[i + 1][0] if i + 1 < len(cuts) else len(text)
                segments.append((text[a:end], b - a))            # (segment, length of its protected marker prefix)
        else:
            segments.append((text, 0))
        # ---- punctuation boundaries inside each segment --------------------------------------------------------------
        out = []
        for seg, keep in segments:
            seg = seg.strip()
            start = 0
            for m in re.finditer(r"[.!?]+[\"'\u201d\u2019)\]]*(?=\s)", seg):
                if m.start() < keep:
                    continue
                nxt = seg[m.end():].lstrip()
                if not nxt or nxt[0].islower():
                    continue
                word = re.search(r"([A-Za-z.]+)$", seg[start:m.start()])
                if word and seg[m.start()] == "." and m.end() - m.start() == 1:
                    abbr = word.group(1).lower().rstrip(".")
                    if abbr in _PREPOSITIVE_ABBREVIATIONS:
                        continue
                    if abbr in _NUMBER_ABBREVIATIONS and nxt[0].isdigit():
                        continue
                    if abbr in _NEVER_FINAL_ABBREVIATIONS:
                        continue
                    if abbr in _STARTER_ABBREVIATIONS and re.match(r"[A-Za-z]+", nxt) and \
                            re.match(r"[A-Za-z]+", nxt).group(0) not in _SENTENCE_STARTERS:
                        continue
                out.append(seg[start:m.end()].strip())
                start = m.end()
            if seg[start:].strip():
                out.append(seg[start:].strip())
        return out

    def save_wav(self, wav, path, pipe_out=None):
        AudioProcessor.save_wav(np.asarray(wav), path, self.output_sample_rate, pipe_out)

    @torch.no_grad()
    def tts_batch(self, sentences, trim=True, speaker_id=None, d_vector=None, language_id=None, durations=None):
        """Synthesize a list of sentences as one padded batch -> list of float32 waveforms (numpy).
        speaker_id / d_vector [D] / language_id apply to every sentence (synthesis.py:166-215 per sentence).
        durations (not a reference feature): per-sentence integer frame counts per token, pinning the models' own
        ceil(exp(logw)) — what a parity harness injects when two fp32 implementations land on different sides of a ceil()."""
        tok = self.tts_model.tokenizer
        ids = [tok.text_to_ids(s) for s in sentences]
        if any(len(i) == 0 for i in ids):
            raise ValueError("a sentence has no symbol of the model's vocabulary")
        T = max(len(i) for i in ids)
        x = torch.zeros(len(ids), T, dtype=torch.int64)
        for r, i in enumerate(ids):
            x[r, : len(i)] = torch.tensor(i)
        xl = torch.tensor([len(i) for i in ids], dtype=torch.int64)
        dev = self.device
        # ragged_exact: every sentence of the padded batch gets the result of a B=1 run on it (reference semantics)
        aux = {"x_lengths": xl.to(dev), "ragged_exact": True}
        if speaker_id is not None:
            aux["speaker_ids"] = torch.full((len(ids),), int(speaker_id), dtype=torch.int64, device=dev)
        if d_vector is not None:
            dv = torch.as_tensor(np.asarray(d_vector), dtype=torch.float32).reshape(1, -1)
            aux["d_vectors"] = dv.expand(len(ids), -1).contiguous().to(dev)
        if language_id is not None:
            aux["language_ids"] = torch.full((len(ids),), int(language_id), dtype=torch.int64, device=dev)
        if durations is not None:
            d = torch.zeros(len(ids), T, dtype=torch.float32)
            for r, dr in enumerate(durations):
                d[r, : len(ids[r])] = torch.as_tensor(dr, dtype=torch.float32).reshape(-1)[: len(ids[r])]
            aux["durations"] = d.to(dev)
        sr_t = _get(_get(self.tts_config, "audio", {}), "sample_rate", 22050)
        sr_v = _get(_get(self.vocoder_config, "audio", {}), "sample_rate", 22050) if self.vocoder_config is not None else sr_t
        do_trim = trim and bool(_get(_get(self.tts_config, "audio", {}), "do_trim_silence", False))
        if self.pipeline is not None and self.pipeline.supported() and sr_t == sr_v:
            self.pipeline.last_ctx = None
            fused = self.pipeline(x.to(dev), aux)           # acoustic model -> seam -> vocoder without leaving the device
            if fused is not None:
                wav, lens = fused
                wav = wav.float().cpu().numpy().reshape(len(ids), -1)
                res = [wav[r, : int(lens[r])] for r in range(len(ids))]
                return [w[: self.tts_model.ap.find_endpoint(w)] for w in res] if do_trim else res
            if self.pipeline.last_ctx is not None:          # a large batch: the front end has run, inference continues from it
                aux = dict(aux, _front_ctx=self.pipeline.last_ctx)
                self.pipeline.last_ctx = None
        out = self.tts_model.inference(x.to(dev), aux)
        frames = out["y_lengths"]
        if self.vocoder_model is None:
            wav = out["model_outputs"]                                        # VITS: [B,1,T_wav]
            lens = (frames * (wav.shape[-1] // out["y_mask"].shape[-1])).tolist()
        else:
            mel = out["model_outputs"].transpose(1, 2)                        # [B,C,T]
            voc_in = mel_renorm_device(mel, self.tts_model.ap, self.vocoder_ap)
            if isinstance(self.tts_model, GlowTTS):                           # squeeze drops an odd last frame
                nsq = self.tts_model.num_squeeze
                frames = torch.div(frames, nsq, rounding_mode="floor") * nsq
            if sr_t != sr_v:
                # interpolate_vocoder_input (synthesizer.py:418-424): bilinear with scale [1, sr_v/sr_t] and
                # recompute_scale_factor=True == linear interpolation along time only
                from . import ops as _ops

                # per sentence, like the reference's sentence loop: each row's valid frames are interpolated on their own
                # and the results re-assembled into one ragged batch for the vocoder
                rows = [_ops.linear_interp(voc_in[r:r + 1, :, : int(frames[r])].contiguous(), sr_v / sr_t,
                                           recompute_scale_factor=True) for r in range(len(ids))]
                frames = torch.tensor([t.shape[2] for t in rows], device=voc_in.device)
                voc_in = torch.zeros((len(ids), voc_in.shape[1], int(frames.max())), dtype=torch.float32, device=voc_in.device)
                for r, t in enumerate(rows):
                    voc_in[r, :, : t.shape[2]] = t[0]
                mel = voc_in
            wav = self.vocoder_model.model_g.inference(voc_in, lengths=frames)
            pad = self.vocoder_model.model_g.inference_padding
            hop_total = wav.shape[-1] // (mel.shape[-1] + 2 * pad)
            lens = ((frames + 2 * pad) * hop_total).tolist()
        wav = wav.float().cpu().numpy().reshape(len(ids), -1)
        res = []
        for r in range(len(ids)):
            w = wav[r, : int(lens[r])]
            if do_trim:
                w = w[: self.tts_model.ap.find_endpoint(w)]
            res.append(w)
        return res

    def tts(self, text="", speaker_name=None, language_name=None, speaker_wav=None, style_wav=None, style_text=None,
            reference_wav=None, reference_speaker_name=None, split_sentences=True, **kwargs):
        """synthesizer.py:257-505: returns a flat list of samples, sentences separated by 10000 zeros (:441).
        Multi-speaker (speaker id table or d-vector file) and multilingual requests resolve names through the model's
        managers with the reference's error behaviour (:301-365); computing a d-vector from `speaker_wav` and
        voice conversion from `reference_wav` need the speaker-encoder network, which is not built."""
        start = time.time()
        if not text:
            raise ValueError("You need to define either `text` (for sythesis) or a `reference_wav` (for voice conversion) to use the Coqui TTS API.")
        if speaker_wav or reference_wav:
            raise _lib.TtsAmdError("speaker_wav / reference_wav requests need the speaker encoder, which is not built")
        sens = self.split_into_sentences(text) if split_sentences else [text]
        spk_mgr = getattr(self.tts_model, "speaker_manager", None)
        speaker_id = d_vector = None
        if self.tts_speakers_file or hasattr(spk_mgr, "name_to_id"):                       # synthesizer.py:307-330
            if speaker_name and isinstance(speaker_name, str):
                if _get(_get(self.tts_config, "model_args", self.tts_config), "use_d_vector_file", False) or \
                        _get(self.tts_config, "use_d_vector_file", False):
                    d_vector = np.array(spk_mgr.get_mean_embedding(speaker_name, num_samples=None, randomize=False))[None, :]
                else:
                    speaker_id = spk_mgr.name_to_id[speaker_name]
            elif len(spk_mgr.name_to_id) == 1:
                speaker_id = list(spk_mgr.name_to_id.values())[0]
            elif not speaker_name:
                raise ValueError(" [!] Looks like you are using a multi-speaker model. You need to define either a "
                                 "`speaker_idx` or a `speaker_wav` to use a multi-speaker model.")
        elif speaker_name and self.voice_dir is None:                                      # :331-336
            raise ValueError(" [!] Missing speakers.json file path for selecting speaker %s."
                             "Define path for speaker.json if it is a multi-speaker model or remove defined speaker idx. "
                             % speaker_name)
        lang_mgr = getattr(self.tts_model, "language_manager", None)
        language_id = None
        if self.tts_languages_file or lang_mgr is not None:                                # :337-365
            if len(lang_mgr.name_to_id) == 1:
                language_id = list(lang_mgr.name_to_id.values())[0]
            elif language_name and isinstance(language_name, str):
                try:
                    language_id = lang_mgr.name_to_id[language_name]
                except KeyError as e:
                    raise ValueError(" [!] Looks like you use a multi-lingual model. Language %s is not in the available "
                                     "languages: %s." % (language_name, lang_mgr.name_to_id.keys())) from e
            elif not language_name:
                raise ValueError(" [!] Look like you use a multi-lingual model. You need to define either a "
                                 "`language_name` or a `style_wav` to use a multi-lingual model.")
        wavs = []
        for w in self.tts_batch(sens, speaker_id=speaker_id, d_vector=d_vector, language_id=language_id):
            wavs.append(w)
            wavs.append(np.zeros(10000, np.float32))
        flat = np.concatenate(wavs)
        dt = time.time() - start
        audio_time = len(flat) / self.output_sample_rate
        print(" > Processing time: %s" % dt)
        print(" > Real-time factor: %s" % (dt / audio_time))
        return flat.tolist()
