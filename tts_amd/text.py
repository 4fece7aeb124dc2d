"""Tokenizer: the part of `TTS.tts.utils.text` that `synthesis()` needs — `TTSTokenizer.text_to_ids` /
`init_from_config` (tokenizer.py:87-134,149-216), the vocabulary classes `Graphemes` (characters.py:280-291,426-466)
and `VitsCharacters` (TTS/tts/models/vits.py:1933-1977), and the regex cleaners of cleaners.py.

`init_from_config` honours `config.characters.characters_class` and `config.text_cleaner` exactly like the reference
(tokenizer.py:159-170): the class / cleaner named by the config is the one used, and anything this build does not
carry RAISES instead of silently falling back to a different vocabulary (a wrong id table is silently wrong speech).
Out of scope: phonemizers (espeak / gruut subprocesses: `use_phonemes=True` raises) and the number /
time expansion of `english_cleaners` / `phoneme_cleaners` (needs the `inflect` package: text containing digits raises).
CPU string processing, as in the reference."""
import re

from .vits import _get

_PAD, _EOS, _BOS, _BLANK = "<PAD>", "<EOS>", "<BOS>", "<BLNK>"
_CHARACTERS = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"
_PUNCTUATIONS = "!'(),-.:;? "
# characters.py:27-35 (default IPA set appended to the graphemes by VitsCharacters)
_PHONEMES = ("iyɨʉɯuɪʏʊeøɘəɵɤoɛœɜɞʌɔæɐaɶɑɒᵻ" "ʘɓǀɗǃʄǂɠǁʛ"
             "pbtdʈɖcɟkɡqɢʔɴŋɲɳnɱmʙrʀⱱɾɽɸβfvθðszʃʒʂʐçʝxɣχʁħʕhɦɬɮʋɹɻjɰlɭʎʟ" "ˈˌːˑ" "ʍwɥʜʢʡɕʑɺɧʲ" "ɚ˞ɫ")


class BaseCharacters:
    """characters.py:118-300: vocab = [pad][eos][bos][blank] + (unique / sorted) characters + punctuations."""

    def __init__(self, characters=None, punctuations=None, pad=None, eos=None, bos=None, blank=None, is_unique=False,
                 is_sorted=True):
        self.characters, self.punctuations = characters, punctuations
        self.pad, self.eos, self.bos, self.blank = pad, eos, bos, blank
        self.is_unique, self.is_sorted = is_unique, is_sorted
        self._create_vocab()

    def _create_vocab(self):
        vocab = self.characters
        if self.is_unique:
            vocab = list(set(vocab))
        if self.is_sorted:
            vocab = sorted(vocab)
        vocab = list(vocab)
        for tok in (self.blank, self.bos, self.eos, self.pad):           # characters.py:287-290 (prepend order)
            if tok is not None and len(tok) > 0:
                vocab = [tok] + vocab
        self.vocab = vocab + list(self.punctuations)
        if self.is_unique:
            assert len(set(self.vocab)) == len(self.vocab), " [!] There are duplicate characters in the character set."

    @property
    def vocab(self):
        return self._vocab

    @vocab.setter
    def vocab(self, v):
        self._vocab = v
        self._char_to_id = {c: i for i, c in enumerate(v)}       # a duplicate keeps its LAST index, as in the reference
        self._id_to_char = dict(enumerate(v))

    @property
    def num_chars(self):
        return len(self._vocab)

    def char_to_id(self, c):
        return self._char_to_id[c]

    def id_to_char(self, i):
        return self._id_to_char[i]

    def _special(self, tok):
        return self._char_to_id[tok] if tok else len(self.vocab)

    pad_id = property(lambda self: self._special(self.pad))
    blank_id = property(lambda self: self._special(self.blank))
    bos_id = property(lambda self: self._special(self.bos))
    eos_id = property(lambda self: self._special(self.eos))


class Graphemes(BaseCharacters):
    def __init__(self, characters=_CHARACTERS, punctuations=_PUNCTUATIONS, pad=_PAD, eos=_EOS, bos=_BOS, blank=_BLANK,
                 is_unique=False, is_sorted=True):
        super().__init__(characters, punctuations, pad, eos, bos, blank, is_unique, is_sorted)

    @staticmethod
    def init_from_config(config):
        """characters.py:469-490: a config with a `characters` block passes it through verbatim (including `None`
        special tokens); without one the class defaults apply."""
        ch = _get(config, "characters", None)
        if ch:
            keys = ("characters", "punctuations", "pad", "eos", "bos", "blank", "is_unique", "is_sorted")
            kw = {k: _get(ch, k) for k in keys if _has(ch, k)}
            return Graphemes(**kw)
        return Graphemes()


class VitsCharacters(BaseCharacters):
    """vits.py:1933-1977: [pad] + punctuations + (graphemes + IPA characters, config order, NOT sorted) + [blank]."""

    def __init__(self, graphemes=_CHARACTERS, punctuations=_PUNCTUATIONS, pad=_PAD, ipa_characters=_PHONEMES):
        if ipa_characters is not None:
            graphemes = graphemes + ipa_characters
        super().__init__(graphemes, punctuations, pad, None, None, "<BLNK>", is_unique=False, is_sorted=True)

    def _create_vocab(self):
        self.vocab = [self.pad] + list(self.punctuations) + list(self.characters) + [self.blank]

    @staticmethod
    def init_from_config(config):
        ch = _get(config, "characters", None)
        if ch is not None:
            return VitsCharacters(graphemes=_get(ch, "characters"), ipa_characters=_get(ch, "phonemes"),
                                  punctuations=_get(ch, "punctuations"), pad=_get(ch, "pad"))
        return VitsCharacters()


_CHARACTER_CLASSES = {
    "TTS.tts.utils.text.characters.Graphemes": Graphemes,
    "TTS.tts.models.vits.VitsCharacters": VitsCharacters,
    "tts_amd.text.Graphemes": Graphemes,
    "tts_amd.text.VitsCharacters": VitsCharacters,
}


def _has(cfg, key):
    return (key in cfg) if isinstance(cfg, dict) else hasattr(cfg, key)


# ---- cleaners (TTS/tts/utils/text/cleaners.py) ---------------------------------------------------------------------
_WS = re.compile(r"\s+")
# english/abbreviations.py
_ABBREVIATIONS_EN = [(re.compile("\\b%s\\." % a, re.IGNORECASE), b) for a, b in [
    ("mrs", "misess"), ("mr", "mister"), ("dr", "doctor"), ("st", "saint"), ("co", "company"), ("jr", "junior"),
    ("maj", "major"), ("gen", "general"), ("drs", "doctors"), ("rev", "reverend"), ("lt", "lieutenant"),
    ("hon", "honorable"), ("sgt", "sergeant"), ("capt", "captain"), ("esq", "esquire"), ("ltd", "limited"),
    ("col", "colonel"), ("ft", "fort")]]


def lowercase(text):
    return text.lower()


def collapse_whitespace(text):
    return re.sub(_WS, " ", text).strip()


def remove_aux_symbols(text):
    return re.sub(r"[\<\>\(\)\[\]\"]+", "", text)


def replace_symbols(text, lang="en"):
    text = text.replace(";", ",")
    text = text.replace("-", " ") if lang != "ca" else text.replace("-", "")
    text = text.replace(":", ",")
    if lang == "en":
        text = text.replace("&", " and ")
    elif lang == "fr":
        text = text.replace("&", " et ")
    elif lang == "pt":
        text = text.replace("&", " e ")
    elif lang == "ca":
        text = text.replace("&", " i ").replace("'", "")
    return text


def expand_abbreviations(text):
    for rx, rep in _ABBREVIATIONS_EN:
        text = re.sub(rx, rep, text)
    return text


def _no_digits(text, who):
    if re.search(r"\d", text):
        raise NotImplementedError("%s: number / time expansion needs the `inflect` package, which is not in this image; "
                                  "spell numbers out or use a cleaner without number normalisation" % who)
    return text


def basic_cleaners(text):
    return collapse_whitespace(lowercase(text))


transliteration_cleaners = basic_cleaners        # the reference's convert_to_ascii step is commented out
basic_german_cleaners = basic_cleaners


def basic_turkish_cleaners(text):
    return collapse_whitespace(lowercase(text.replace("I", "ı")))


def english_cleaners(text):
    text = _no_digits(lowercase(text), "english_cleaners")       # expand_time_english / en_normalize_numbers: digits only
    return collapse_whitespace(remove_aux_symbols(replace_symbols(expand_abbreviations(text))))


def phoneme_cleaners(text):
    text = _no_digits(text, "phoneme_cleaners")
    return collapse_whitespace(remove_aux_symbols(replace_symbols(expand_abbreviations(text))))


def portuguese_cleaners(text):
    return collapse_whitespace(remove_aux_symbols(replace_symbols(lowercase(text), lang="pt")))


def multilingual_cleaners(text):
    return collapse_whitespace(remove_aux_symbols(replace_symbols(lowercase(text), lang=None)))


def no_cleaners(text):
    return text.replace("\n", "")


_CLEANERS = {f.__name__ if f.__name__ != "basic_cleaners" else "basic_cleaners": f for f in (
    basic_cleaners, basic_turkish_cleaners, english_cleaners, phoneme_cleaners, portuguese_cleaners,
    multilingual_cleaners, no_cleaners)}
_CLEANERS.update(transliteration_cleaners=transliteration_cleaners, basic_german_cleaners=basic_german_cleaners)


def get_cleaner(name):
    """tokenizer.py:159-161 `getattr(cleaners, config.text_cleaner)`; cleaners this build does not carry (french — needs
    its abbreviation table —, chinese_mandarin) raise."""
    if name not in _CLEANERS:
        raise NotImplementedError("text_cleaner %r is not implemented in tts_amd.text (have: %s)" % (name, sorted(_CLEANERS)))
    return _CLEANERS[name]


class TTSTokenizer:
    def __init__(self, use_phonemes=False, text_cleaner=None, characters=None, add_blank=False, use_eos_bos=False):
        if use_phonemes:
            raise NotImplementedError("phonemizers (espeak/gruut) are outside this build's scope; use a grapheme model")
        self.text_cleaner = text_cleaner
        self.characters = characters or Graphemes()
        self.add_blank, self.use_eos_bos = add_blank, use_eos_bos
        self.not_found_characters = []

    @staticmethod
    def init_from_config(config, characters=None):
        """tokenizer.py:149-216: cleaner by name, character class by `characters.characters_class` (else Graphemes for
        grapheme models); returns (tokenizer, config)."""
        cleaner_name = _get(config, "text_cleaner", None)
        text_cleaner = get_cleaner(cleaner_name) if isinstance(cleaner_name, str) and cleaner_name else None
        if characters is None:
            ch = _get(config, "characters", None)
            cls_path = _get(ch, "characters_class", None) if ch else None
            if cls_path:
                if cls_path not in _CHARACTER_CLASSES:
                    raise NotImplementedError("characters_class %r is not implemented in tts_amd.text (have: %s)"
                                              % (cls_path, sorted(_CHARACTER_CLASSES)))
                characters = _CHARACTER_CLASSES[cls_path].init_from_config(config)
            elif bool(_get(config, "use_phonemes", False)):
                raise NotImplementedError("IPAPhonemes / phonemizers are outside this build's scope")
            else:
                characters = Graphemes.init_from_config(config)
        tok = TTSTokenizer(bool(_get(config, "use_phonemes", False)), text_cleaner, characters,
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

    def decode(self, ids):
        return "".join(self.characters.id_to_char(i) for i in ids)

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

    def ids_to_text(self, ids):
        return self.decode(ids)
