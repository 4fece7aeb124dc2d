"""Speaker / language look-up tables used by the multi-speaker and multilingual request path
(`Synthesizer.tts(speaker_name=..., language_name=...)`, synthesizer.py:301-365): host-side mirrors of
`TTS.tts.utils.speakers.SpeakerManager` (speakers.py:14-117 over managers.py:36-300) and
`TTS.tts.utils.languages.LanguageManager` (languages.py:13-100) restricted to what inference reads — the name -> id
maps, the d-vector file (`{clip: {"name": speaker, "embedding": [...]}}`) and the per-speaker mean embedding.
The speaker-encoder network that turns a reference clip into a d-vector is outside this build."""
import json

import numpy as np
import torch


def _get(cfg, key, default=None):
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def _from_config_or_model_args(config, key, default=None):
    """generic_utils.get_from_config_or_model_args_with_default: model_args first, then the top level."""
    margs = _get(config, "model_args", None)
    v = _get(margs, key, None)
    if v is None:
        v = _get(config, key, None)
    return default if v is None else v


def load_file(path):
    """managers.py:14-22: .json -> json, anything else -> torch.load."""
    if str(path).endswith(".json"):
        with open(path, "r") as f:
            return json.load(f)
    return torch.load(path, map_location="cpu", weights_only=False)


class BaseIDManager:
    def __init__(self, id_file_path=""):
        self.name_to_id = {}
        if id_file_path:
            self.load_ids_from_file(id_file_path)

    def load_ids_from_file(self, file_path):
        self.name_to_id = load_file(file_path)


class SpeakerManager(BaseIDManager):
    def __init__(self, d_vectors_file_path="", speaker_id_file_path=""):
        super().__init__(speaker_id_file_path)
        self.embeddings, self.embeddings_by_names, self.clip_ids = {}, {}, []
        self.encoder = self.encoder_ap = None
        files = d_vectors_file_path if isinstance(d_vectors_file_path, (list, tuple)) else [d_vectors_file_path]
        files = [f for f in files if f]
        if files:
            self._load_embedding_files(files, renumber=isinstance(d_vectors_file_path, (list, tuple)))

    def _load_embedding_files(self, paths, renumber):
        names = {}
        for path in paths:
            emb = load_file(path)
            dup = set(self.embeddings) & set(emb)
            if dup:
                raise ValueError(" [!] Duplicate embedding names <%s> in %s" % (dup, path))
            for i, name in enumerate(sorted({v["name"] for v in emb.values()})):   # managers.py:195-196
                names[name] = i
            for clip, v in emb.items():
                self.embeddings_by_names.setdefault(v["name"], []).append(v["embedding"])
            self.clip_ids.extend(emb.keys())
            self.embeddings.update(emb)
        # a list of files is re-numbered in first-seen order (managers.py:238-239); a single file keeps sorted order
        self.name_to_id = {n: i for i, n in enumerate(names)} if renumber else names

    @property
    def num_speakers(self):
        return len(self.name_to_id)

    @property
    def speaker_names(self):
        return list(self.name_to_id.keys())

    @property
    def embedding_dim(self):
        if self.embeddings:
            return len(next(iter(self.embeddings.values()))["embedding"])
        return 0

    def get_speakers(self):
        return self.name_to_id

    def get_embedding_by_clip(self, clip_idx):
        return self.embeddings[clip_idx]["embedding"]

    def get_embeddings_by_name(self, idx):
        return self.embeddings_by_names[idx]

    def get_mean_embedding(self, idx, num_samples=None, randomize=False):
        """managers.py:273-293 (randomize=True draws with replacement there; inference never asks for it)."""
        emb = self.get_embeddings_by_name(idx)
        if num_samples is not None:
            assert len(emb) >= num_samples, " [!] %s has number of samples < %d" % (idx, num_samples)
            if randomize:
                raise NotImplementedError("random sub-sampling of d-vectors is a training-time feature")
            emb = emb[:num_samples]
        return np.stack([np.asarray(e) for e in emb]).mean(0)

    @staticmethod
    def init_from_config(config, samples=None):
        """speakers.py:86-117 without the dataset-driven branch."""
        mgr = None
        if _from_config_or_model_args(config, "use_speaker_embedding", False):
            for key in ("speaker_file", "speakers_file"):
                path = _from_config_or_model_args(config, key, None)
                if path:
                    mgr = SpeakerManager(speaker_id_file_path=path)
        if _from_config_or_model_args(config, "use_d_vector_file", False):
            mgr = SpeakerManager()
            path = _from_config_or_model_args(config, "d_vector_file", None)
            if path:
                mgr = SpeakerManager(d_vectors_file_path=path)
        return mgr


class LanguageManager(BaseIDManager):
    def __init__(self, language_ids_file_path="", config=None):
        super().__init__(language_ids_file_path)
        if config is not None:
            self.name_to_id = self.parse_language_ids_from_config(config)

    @property
    def num_languages(self):
        return len(self.name_to_id)

    @property
    def language_names(self):
        return list(self.name_to_id.keys())

    @staticmethod
    def parse_language_ids_from_config(c):
        """languages.py:47-62: sorted set of the datasets' `language` fields."""
        langs = set()
        for ds in _get(c, "datasets", None) or []:
            lang = _get(ds, "language", None)
            if not lang:
                raise ValueError("Dataset %s has no language specified." % _get(ds, "name", "?"))
            langs.add(lang)
        return {name: i for i, name in enumerate(sorted(langs))}

    @staticmethod
    def init_from_config(config):
        """languages.py:88-100.  (The reference builds the file-backed manager and then unconditionally replaces it by
        the datasets-derived one; a deployment config has no datasets, so the file wins here when it is given.)"""
        if not _from_config_or_model_args(config, "use_language_embedding", False):
            return None
        path = _from_config_or_model_args(config, "language_ids_file", None)
        if path:
            return LanguageManager(language_ids_file_path=path)
        return LanguageManager(config=config)
