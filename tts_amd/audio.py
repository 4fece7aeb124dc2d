"""The slice of `TTS.utils.audio.AudioProcessor` that sits on the inference path: spectrogram
(de)normalisation at the acoustic-model -> vocoder seam (processor.py:259-336), silence trimming of the
synthesised waveform (numpy_transforms.py:326-377) and `save_wav` (numpy_transforms.py:430-447).  Feature
extraction (STFT / mel / Griffin-Lim) is training-side and not built.

`normalize` / `denormalize` are the reference's numpy arithmetic (host, used by callers that hold numpy mels);
`mel_renorm_device` runs the composed `vocoder_ap.normalize(tts_ap.denormalize(.))` as one HIP kernel on the
device tensor, removing the D2H -> numpy -> H2D hop of synthesizer.py:412-429.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import P, check, lib, stream_ptr
from .vits import _get

AUDIO_DEFAULTS = dict(  # BaseAudioConfig, TTS/config/shared_configs.py:100-160
    sample_rate=22050, num_mels=80, hop_length=256, win_length=1024, fft_size=1024, signal_norm=True,
    symmetric_norm=True, max_norm=4.0, min_level_db=-100, ref_level_db=20, clip_norm=True, do_trim_silence=True,
    trim_db=45, stats_path=None)


class MelNorm(ctypes.Structure):
    """Mirror of `ttsamd_mel_norm` (include/tts_amd.h)."""

    _fields_ = [("signal_norm", ctypes.c_int32), ("symmetric_norm", ctypes.c_int32), ("clip_norm", ctypes.c_int32),
                ("max_norm", ctypes.c_float), ("min_level_db", ctypes.c_float), ("ref_level_db", ctypes.c_float),
                ("mean", ctypes.c_void_p), ("scale", ctypes.c_void_p)]


class AudioProcessor:
    def __init__(self, **kwargs):
        for k, v in AUDIO_DEFAULTS.items():
            setattr(self, k, kwargs.get(k, v))
        self.mel_mean = self.mel_scale = None
        if self.stats_path and self.signal_norm:     # processor.py ctor tail + setup_scaler (:367-381)
            stats = np.load(self.stats_path, allow_pickle=True).item()
            self.mel_mean = np.asarray(stats["mel_mean"], np.float32)
            self.mel_scale = np.asarray(stats["mel_std"], np.float32)
            self.max_norm = self.clip_norm = self.symmetric_norm = None
        self._dev = {}

    @staticmethod
    def init_from_config(config, verbose=True):
        audio = _get(config, "audio", config)
        kw = {k: _get(audio, k) for k in AUDIO_DEFAULTS if _get(audio, k) is not None}
        return AudioProcessor(**kw)

    # ---- numpy path (reference arithmetic) --------------------------------------------------------
    def normalize(self, S):
        S = S.copy()
        if not self.signal_norm:
            return S
        if self.mel_mean is not None:
            return ((S.T - self.mel_mean) / self.mel_scale).T
        S -= self.ref_level_db
        S_norm = (S - self.min_level_db) / (-self.min_level_db)
        if self.symmetric_norm:
            S_norm = ((2 * self.max_norm) * S_norm) - self.max_norm
            return np.clip(S_norm, -self.max_norm, self.max_norm) if self.clip_norm else S_norm
        S_norm = self.max_norm * S_norm
        return np.clip(S_norm, 0, self.max_norm) if self.clip_norm else S_norm

    def denormalize(self, S):
        S_denorm = S.copy()
        if not self.signal_norm:
            return S_denorm
        if self.mel_mean is not None:
            return (S_denorm.T * self.mel_scale + self.mel_mean).T
        if self.symmetric_norm:
            if self.clip_norm:
                S_denorm = np.clip(S_denorm, -self.max_norm, self.max_norm)
            S_denorm = ((S_denorm + self.max_norm) * -self.min_level_db / (2 * self.max_norm)) + self.min_level_db
            return S_denorm + self.ref_level_db
        if self.clip_norm:
            S_denorm = np.clip(S_denorm, 0, self.max_norm)
        S_denorm = (S_denorm * -self.min_level_db / self.max_norm) + self.min_level_db
        return S_denorm + self.ref_level_db

    # ---- waveform post-processing (host; the reference's numpy code) --------------------------------
    def find_endpoint(self, wav, min_silence_sec=0.8):
        window_length = int(self.sample_rate * min_silence_sec)
        hop_length = int(window_length / 4)
        threshold = 10 ** (-self.trim_db / 20.0)           # db_to_amp(x=-trim_db, gain=20, base=10)
        for x in range(hop_length, len(wav) - window_length, hop_length):
            if np.max(wav[x: x + window_length]) < threshold:
                return x + hop_length
        return len(wav)

    @staticmethod
    def save_wav(wav, path, sample_rate, pipe_out=None):
        import scipy.io.wavfile

        wav = np.asarray(wav, dtype=np.float32)
        wav_norm = wav * (32767 / max(0.01, np.max(np.abs(wav))))
        scipy.io.wavfile.write(path, sample_rate, wav_norm.astype(np.int16))

    # ---- device seam ----------------------------------------------------------------------------------
    def _mel_norm_struct(self, device):
        s = MelNorm()
        s.signal_norm = int(bool(self.signal_norm))
        s.symmetric_norm, s.clip_norm = int(bool(self.symmetric_norm)), int(bool(self.clip_norm))
        s.max_norm = float(self.max_norm or 0.0)
        s.min_level_db, s.ref_level_db = float(self.min_level_db), float(self.ref_level_db)
        if self.mel_mean is not None:
            key = str(device)
            if key not in self._dev:
                self._dev[key] = (torch.from_numpy(self.mel_mean).to(device), torch.from_numpy(self.mel_scale).to(device))
            s.mean, s.scale = self._dev[key][0].data_ptr(), self._dev[key][1].data_ptr()
        return s


def mel_renorm_device(mel, tts_ap, vocoder_ap):
    """mel [B,C,T] on the GPU -> vocoder_ap.normalize(tts_ap.denormalize(mel)) [B,C,T] (include/tts_amd.h)."""
    _lib.require_gpu(mel, "mel")
    mel = mel.float().contiguous()
    B, C, T = mel.shape
    y = torch.empty_like(mel)
    a, b = tts_ap._mel_norm_struct(mel.device), vocoder_ap._mel_norm_struct(mel.device)
    check(lib().ttsamd_mel_renorm(P(y), P(mel), ctypes.byref(a), ctypes.byref(b), B, C, T, stream_ptr()), "mel_renorm")
    return y
