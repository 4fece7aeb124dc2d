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


def test_host_side_policy_queries_need_no_gpu():
    """The pure-host queries of the ABI (no kernel behind them): version, which (kernel, dilation) pairs have a tuned conv
    instantiation vs the generic kernel, which fused pairs exist, and the shape window of the grouped MRF launch."""
    L = ctypes.CDLL(_lib.LIB_PATH)
    assert L.ttsamd_abi_version() == 4
    for k, d, sup, tuned in ((11, 1, 1, 1), (3, 9, 1, 1), (9, 2, 1, 0), (31, 27, 1, 0), (32, 1, 0, 0), (3, 28, 0, 0), (4, 1, 1, 0)):
        assert (L.ttsamd_conv1d_supported(k, d), L.ttsamd_conv1d_tuned(k, d)) == (sup, tuned), (k, d)
    assert L.ttsamd_resblock_pair_supported(16, 7, 3) == 1 and L.ttsamd_resblock_pair_supported(48, 7, 3) == 0
    L.ttsamd_resblock_weight_bytes.restype = ctypes.c_size_t
    L.ttsamd_conv1d_packed_split_bytes.restype = ctypes.c_size_t
    assert L.ttsamd_resblock_weight_bytes(8, 3) == L.ttsamd_conv1d_packed_split_bytes(32, 32, 3)      # zero-padded to the 32-channel tile
    # grouped MRF launch: a single sentence's stages yes, one utterance's / a batch's no
    for c, t, b, want in ((64, 2624, 1, 1), (32, 20992, 1, 1), (16, 41984, 1, 1), (8, 83968, 1, 1), (64, 98560, 1, 0), (32, 197120, 1, 0),
                          (128, 1000, 1, 0), (32, 5000, 8, 0)):
        assert L.ttsamd_resblock_group_supported(c, t, b) == want, (c, t, b)
