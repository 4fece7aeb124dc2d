"""`TTS.api.TTS`-shaped entry point (TTS/api.py:14-343) for locally stored checkpoints.

`TTS(model_path=..., config_path=..., vocoder_path=..., vocoder_config_path=..., gpu=True)` then
`tts(text)` / `tts_to_file(text, file_path=...)` — same argument names as the reference.  `model_name=` (model-zoo
download, api.py:128-137) needs the network and the ModelManager, both out of scope: it raises."""
from . import _lib
from .synthesizer import Synthesizer


class TTS:
    def __init__(self, model_name="", model_path=None, config_path=None, vocoder_path=None, vocoder_config_path=None,
                 progress_bar=True, gpu=True):
        if model_name:
            raise _lib.TtsAmdError("model_name= needs the model zoo / network (out of scope): pass model_path/config_path")
        if not model_path and not config_path:
            raise ValueError("model_path / config_path are required")
        self.synthesizer = Synthesizer(tts_checkpoint=model_path or "", tts_config_path=config_path,
                                       vocoder_checkpoint=vocoder_path or "", vocoder_config=vocoder_config_path or "",
                                       use_cuda=gpu)

    @property
    def is_multi_speaker(self):
        """api.py:84-93: the loaded model's speaker manager decides."""
        m = getattr(self.synthesizer.tts_model, "speaker_manager", None)
        return m is not None and m.num_speakers > 1

    @property
    def is_multi_lingual(self):
        """api.py:95-107 (the XTTS / config.languages clauses concern models this build does not carry)."""
        m = getattr(self.synthesizer.tts_model, "language_manager", None)
        return m is not None and m.num_languages > 1

    @property
    def speakers(self):
        return self.synthesizer.tts_model.speaker_manager.speaker_names if self.is_multi_speaker else None

    @property
    def languages(self):
        return self.synthesizer.tts_model.language_manager.language_names if self.is_multi_lingual else None

    def _check_arguments(self, speaker=None, language=None, speaker_wav=None, emotion=None, speed=None, **kwargs):
        """api.py:215-235, same messages."""
        if self.is_multi_speaker and (speaker is None and speaker_wav is None):
            raise ValueError("Model is multi-speaker but no `speaker` is provided.")
        if self.is_multi_lingual and language is None:
            raise ValueError("Model is multi-lingual but no `language` is provided.")
        if not self.is_multi_speaker and speaker is not None and "voice_dir" not in kwargs:
            raise ValueError("Model is not multi-speaker but `speaker` is provided.")
        if not self.is_multi_lingual and language is not None:
            raise ValueError("Model is not multi-lingual but `language` is provided.")
        if emotion is not None and speed is not None:
            raise ValueError("Emotion and speed can only be used with Coqui Studio models. Which is discontinued.")

    def tts(self, text, speaker=None, language=None, speaker_wav=None, emotion=None, speed=None, split_sentences=True,
            **kwargs):
        """api.py:237-288: names go to the Synthesizer as speaker_name / language_name."""
        self._check_arguments(speaker=speaker, language=language, speaker_wav=speaker_wav, emotion=emotion, speed=speed,
                              **kwargs)
        return self.synthesizer.tts(text=text, speaker_name=speaker, language_name=language, speaker_wav=speaker_wav,
                                    reference_wav=None, style_wav=None, style_text=None, reference_speaker_name=None,
                                    split_sentences=split_sentences, **kwargs)

    def tts_to_file(self, text, speaker=None, language=None, speaker_wav=None, emotion=None, speed=1.0, pipe_out=None,
                    file_path="output.wav", split_sentences=True, **kwargs):
        wav = self.tts(text=text, speaker=speaker, language=language, speaker_wav=speaker_wav,
                       split_sentences=split_sentences, **kwargs)
        self.synthesizer.save_wav(wav=wav, path=file_path, pipe_out=pipe_out)
        return file_path
