"""TEST INFRASTRUCTURE ONLY — import shim for the *real* reference leaf modules.

Only usable in the build container where /root/reference exists (it is absent on
the GPU box).  Used by tests/ (when the tree is present) and by
tests/golden/make_golden.py to pin oracle/tts_oracle.py against the reference's
own nn.Modules.  Never imported by the product package `tts_amd`.

Recipe: stub `coqpit`, register an empty `TTS.tts.layers`
package so `TTS/tts/layers/__init__.py` (-> losses -> librosa) is skipped, then
import the leaf modules straight from /root/reference.
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("TTS_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "TTS"))


_done = False


def install():
    global _done
    if _done:
        return
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    if "coqpit" not in sys.modules:
        stub = types.ModuleType("coqpit")
        stub.Coqpit = type("Coqpit", (), {})
        stub.check_argument = lambda *a, **k: None
        sys.modules["coqpit"] = stub
    if "torchaudio" not in sys.modules:
        try:
            import torchaudio  # noqa: F401
        except Exception:  # only the XTTS speaker encoder touches it (never instantiated here)
            sys.modules["torchaudio"] = types.ModuleType("torchaudio")
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import TTS  # noqa: F401
    import TTS.tts  # noqa: F401

    if "TTS.tts.layers" not in sys.modules:
        pkg = types.ModuleType("TTS.tts.layers")
        pkg.__path__ = [os.path.join(REF_ROOT, "TTS", "tts", "layers")]
        sys.modules["TTS.tts.layers"] = pkg
    _done = True


def ref(module: str):
    """Import a reference leaf module, e.g. ref('TTS.vocoder.models.hifigan_generator')."""
    install()
    return importlib.import_module(module)
