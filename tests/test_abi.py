"""CPU-side check: libtts_amd.so loads and exports every symbol include/tts_amd.h declares."""
import ctypes
import os

from tts_amd import _lib


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g

    g.build()
    assert os.path.exists(_lib.LIB_PATH)
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _lib.declared_symbols()
    assert len(names) >= 5
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    L.ttsamd_arch.restype = ctypes.c_char_p
    assert L.ttsamd_arch() == b"gfx950"


def test_missing_gpu_fails_loudly():
    import pytest
    import torch

    from tts_amd import helpers

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.TtsAmdError):
        helpers.maximum_path(torch.zeros(1, 2, 3), torch.ones(1, 2, 3))
