"""XTTS streaming chunker — the vocoder half of `Xtts.inference_stream` (TTS/tts/models/xtts.py:585-692).

The reference keeps every GPT latent produced so far, re-vocodes the WHOLE prefix each time `stream_chunk_size` new
tokens have arrived, and emits `wav_gen[len(prev) - overlap : -overlap]` with a linear cross-fade over the previous
chunk's last `overlap` samples (`handle_chunks`, xtts.py:585-607).  That is O(n²) vocoder work per sentence.

The generator is a finite-receptive-field conv stack, and the HIP conv kernel accumulates every output sample in an
order that does not depend on where the tile sits, so re-vocoding only the tail — the samples that can still change or
have not been emitted yet, plus `context_frames()` of left context — gives the same bits as re-vocoding the prefix.
`XttsStreamer(windowed=True)` (default) does that: O(n) work, constant per-chunk latency;  `windowed=False` is the
reference's literal schedule.  tests/test_hifigan_gpu.py checks the two bit-for-bit and both against the oracle.

The GPT-2 acoustic model that produces the latents is outside this build (parity unpinned): latents are an
input — any iterable yielding `[C]` / `[n, C]` tensors, e.g. `(latent for _, latent in gpt_generator)`."""
import math

import torch

from . import _lib, ops


def _crossfade_tail(tail, first, wav_overlap, overlap_len):
    """`tail` = wav_gen[a:], a = 0 for the first chunk else len(wav_gen_prev) - overlap_len.  Returns
    (wav_chunk, next wav_overlap) exactly as xtts.py:585-607 would from the full wav_gen."""
    n = tail.shape[0]
    cut = max(0, n - overlap_len)
    body = tail[:cut]
    if wav_overlap is not None:
        if overlap_len > body.shape[0]:
            # chunk shorter than the overlap (last chunk of a sentence): hand over everything that is left, no fade
            return (tail if not first else tail[cut:]), None
        fade_in = torch.linspace(0.0, 1.0, overlap_len, device=tail.device)
        fade_out = torch.linspace(1.0, 0.0, overlap_len, device=tail.device)
        head = body[:overlap_len] * fade_in
        body[:overlap_len] = wav_overlap * fade_out
        body[:overlap_len] += head
    return body, tail[cut:]


def handle_chunks(wav_gen, wav_gen_prev, wav_overlap, overlap_len):
    """Mirror of `Xtts.handle_chunks` (xtts.py:585-607): same arguments, same (wav_chunk, wav_gen_prev, wav_overlap)
    result, same in-place cross-fade into `wav_gen`."""
    first = wav_gen_prev is None
    a = 0 if first else wav_gen_prev.shape[0] - overlap_len
    chunk, overlap = _crossfade_tail(wav_gen[a:], first, wav_overlap, overlap_len)
    return chunk, wav_gen, overlap


class XttsStreamer:
    """decoder: `tts_amd.xtts_decoder.HifiDecoder`.  `stream(latents, g)` yields waveform chunks [n] on the device."""

    def __init__(self, decoder, stream_chunk_size=20, overlap_wav_len=1024, length_scale=1.0, windowed=True):
        self.decoder = decoder
        self.stream_chunk_size = int(stream_chunk_size)
        self.overlap_wav_len = int(overlap_wav_len)
        self.length_scale = float(length_scale)
        self.windowed = bool(windowed)
        self.frames_decoded = 0       # generator input frames vocoded so far (the O(n) vs O(n^2) bookkeeping)

    def _features(self, latents):
        """[n, C] latents -> generator input z [1, C, F] (xtts.py:673-677 + hifigan_decoder.py:686-697)."""
        z = latents.float().t().contiguous()[None]
        if self.length_scale != 1.0:
            z = ops.linear_interp(z, self.length_scale)
        d = self.decoder
        z = ops.linear_interp(z, d.ar_mel_length_compression / d.output_hop_length)
        if d.output_sample_rate != d.input_sample_rate:
            z = ops.linear_interp(z, d.output_sample_rate / d.input_sample_rate)
        return z

    def stream(self, latents, g):
        gen = self.decoder.waveform_decoder
        hop = gen.hop_length()
        ctx = gen.context_frames()
        ov = self.overlap_wav_len
        have, pending = [], 0
        prev_len, wav_overlap = None, None
        it = iter(latents)
        done = False
        while not done:
            try:
                lat = next(it)
                lat = lat.reshape(-1, lat.shape[-1])
                _lib.require_gpu(lat, "latents")
                have.append(lat)
                pending += lat.shape[0]
            except StopIteration:
                done = True
            if not (done or (self.stream_chunk_size > 0 and pending >= self.stream_chunk_size)):
                continue
            if not have:
                return
            z = self._features(torch.cat(have, 0))
            total = z.shape[2] * hop
            first = prev_len is None
            a = 0 if first else prev_len - ov
            f0 = max(0, a // hop - ctx) if self.windowed else 0
            wav = gen.forward(z[:, :, f0:].contiguous(), g=g).reshape(-1)
            self.frames_decoded += z.shape[2] - f0
            chunk, wav_overlap = _crossfade_tail(wav[a - f0 * hop:], first, wav_overlap, ov)
            prev_len, pending = total, 0
            yield chunk

    __call__ = stream


def context_frames(upsample_factors, resblock_type, resblock_kernel_sizes, resblock_dilation_sizes, pre_kernel=7,
                   post_kernel=7):
    """Upper bound, in generator input frames, of how far one output sample of a HiFiGAN generator
    (hifigan_generator.py:162-265) looks to either side."""
    reach = (pre_kernel - 1) / 2.0
    rate = 1
    for u in upsample_factors:
        reach += 1.0 / rate                     # ConvTranspose1d k = 2u, stride u: one input step either side
        rate *= u
        rb = 0
        for k, dils in zip(resblock_kernel_sizes, resblock_dilation_sizes):
            h = (k - 1) // 2
            r = sum(h * d + (h if str(resblock_type) == "1" else 0) for d in dils)
            rb = max(rb, r)
        reach += rb / rate
    reach += (post_kernel - 1) / 2.0 / rate
    return int(math.ceil(reach)) + 1
