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
        return False

    @property
    def is_multi_lingual(self):
        return False

    def tts(self, text, speaker=None, language=None, speaker_wav=None, emotion=None, speed=None, split_sentences=True,
            **kwargs):
        if speaker or language or speaker_wav:
            raise ValueError("Model is not multi-speaker / multi-lingual.")     # api.py:215-235 _check_arguments
        return self.synthesizer.tts(text=text, split_sentences=split_sentences, **kwargs)

    def tts_to_file(self, text, speaker=None, language=None, speaker_wav=None, emotion=None, speed=1.0, pipe_out=None,
                    file_path="output.wav", split_sentences=True, **kwargs):
        wav = self.tts(text=text, speaker=speaker, language=language, speaker_wav=speaker_wav,
                       split_sentences=split_sentences, **kwargs)
        self.synthesizer.save_wav(wav=wav, path=file_path, pipe_out=pipe_out)
        return file_path
