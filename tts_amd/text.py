"""Grapheme tokenizer: the part of `TTS.tts.utils.text` that `synthesis()` needs
(`TTSTokenizer.text_to_ids`, tokenizer.py:87-134; `Graphemes` vocabulary, characters.py:280-291,426-466).
Phonemizers (espeak / gruut subprocesses) are out of scope — a config with `use_phonemes=True`
raises.  CPU string processing, exactly as in the reference."""
import re

from .vits import _get

_PAD, _EOS, _BOS, _BLANK = "<PAD>", "<EOS>", "<BOS>", "<BLNK>"
_CHARACTERS = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"
_PUNCTUATIONS = "!'(),-.:;? "


class Graphemes:
    def __init__(self, characters=_CHARACTERS, punctuations=_PUNCTUATIONS, pad=_PAD, eos=_EOS, bos=_BOS, blank=_BLANK,
                 is_unique=False, is_sorted=True):
        vocab = list(characters)
        if is_unique:
            vocab = list(set(vocab))
        if is_sorted:
            vocab = sorted(vocab)
        for tok in (blank, bos, eos, pad):                       # characters.py:287-290 (prepend order)
            if tok is not None and len(tok) > 0:
                vocab = [tok] + vocab
        self.vocab = vocab + list(punctuations)
        self.pad, self.eos, self.bos, self.blank = pad, eos, bos, blank
        self._char_to_id = {c: i for i, c in enumerate(self.vocab)}

    @property
    def num_chars(self):
        return len(self.vocab)

    def char_to_id(self, c):
        return self._char_to_id[c]

    def _special(self, tok):
        return self._char_to_id[tok] if tok else len(self.vocab)

    pad_id = property(lambda self: self._special(self.pad))
    blank_id = property(lambda self: self._special(self.blank))
    bos_id = property(lambda self: self._special(self.bos))
    eos_id = property(lambda self: self._special(self.eos))


def basic_cleaners(text):
    """TTS/tts/utils/text/cleaners.py `basic_cleaners`: lowercase + collapse whitespace."""
    return re.sub(r"\s+", " ", text.lower()).strip()


class TTSTokenizer:
    def __init__(self, use_phonemes=False, text_cleaner=basic_cleaners, characters=None, add_blank=False,
                 use_eos_bos=False):
        if use_phonemes:
            raise NotImplementedError("phonemizers (espeak/gruut) are outside this build's scope; use a grapheme model")
        self.text_cleaner = text_cleaner
        self.characters = characters or Graphemes()
        self.add_blank, self.use_eos_bos = add_blank, use_eos_bos
        self.not_found_characters = []

    @staticmethod
    def init_from_config(config):
        ch = _get(config, "characters", None)
        chars = Graphemes(**{k: _get(ch, k) for k in ("characters", "punctuations", "pad", "eos", "bos", "blank",
                                                      "is_unique", "is_sorted") if _get(ch, k) is not None}) \
            if ch else Graphemes()
        tok = TTSTokenizer(bool(_get(config, "use_phonemes", False)), basic_cleaners, chars,
                           bool(_get(config, "add_blank", False)), bool(_get(config, "enable_eos_bos_chars", False)))
        return tok, config

    def encode(self, text):
        ids = []
        for c in text:
            if c in self.characters._char_to_id:
                ids.append(self.characters._char_to_id[c])
            elif c not in self.not_found_characters:                # discard but remember (tokenizer.py:72-77)
                self.not_found_characters.append(c)
        return ids

    def text_to_ids(self, text, language=None):
        if self.text_cleaner is not None:
            text = self.text_cleaner(text)
        ids = self.encode(text)
        if self.add_blank:                                           # tokenizer.py:126-134
            out = [self.characters.blank_id] * (len(ids) * 2 + 1)
            out[1::2] = ids
            ids = out
        if self.use_eos_bos:
            ids = [self.characters.bos_id] + list(ids) + [self.characters.eos_id]
        return ids
