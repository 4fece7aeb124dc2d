"""TEST INFRASTRUCTURE ONLY — MAS / alignment-helper oracles (numpy + C).

Restates:
  * TTS/tts/utils/helpers.py:43-57    sequence_mask
  * TTS/tts/utils/helpers.py:154-169  generate_path
  * TTS/tts/utils/helpers.py:172-194  maximum_path / maximum_path_cython (wrapper glue)
  * TTS/tts/utils/helpers.py:197-236  maximum_path_numpy
  * TTS/tts/utils/monotonic_align/core.pyx:11-47 via oracle/mas_oracle.c (C restatement)
and exposes the reference's own compiled Cython core from oracle/_ref/ when it was built.

Pinning: tests/test_oracle_mas.py checks C-restatement == reference Cython == numpy
restatement bit-for-bit, and generate_path against the structure the reference's own test
asserts (tests/tts_tests/test_helpers.py:71-88).
"""
import ctypes
import glob
import importlib.util
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def build():
    """Compile the C restatement (and oracle/_ref when /root/reference is present)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def _load():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libmas_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", _HERE, "libmas_oracle.so"])
        _lib = ctypes.CDLL(path)
        _lib.mas_oracle_c.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float]
        _lib.mas_oracle_c.restype = None
    return _lib


def maximum_path_c(paths, values, t_xs, t_ys, max_neg_val=-1e9):
    """Same signature/semantics as core.pyx:42 maximum_path_c (in place on `values`, `paths` pre-zeroed)."""
    assert paths.dtype == np.int32 and values.dtype == np.float32
    assert paths.flags.c_contiguous and values.flags.c_contiguous
    t_xs = np.ascontiguousarray(t_xs, dtype=np.int32)
    t_ys = np.ascontiguousarray(t_ys, dtype=np.int32)
    b, tx, ty = values.shape
    _load().mas_oracle_c(paths.ctypes.data, values.ctypes.data, t_xs.ctypes.data, t_ys.ctypes.data,
                         b, tx, ty, ctypes.c_float(max_neg_val))


def ref_maximum_path_c():
    """The reference's own Cython `maximum_path_c` built into oracle/_ref/, or None."""
    cands = glob.glob(os.path.join(_HERE, "_ref", "ref_mas_core*.so"))
    if not cands:
        return None
    spec = importlib.util.spec_from_file_location("ref_mas_core", cands[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.maximum_path_c


def sequence_mask(lengths, max_len=None):
    """helpers.py:43-57."""
    lengths = np.asarray(lengths)
    if max_len is None:
        max_len = int(lengths.max())
    return np.arange(max_len)[None, :] < lengths[:, None]


def generate_path(duration, mask):
    """helpers.py:154-169: duration [B,Tx] (integer-valued), mask [B,Tx,Ty] -> path [B,Tx,Ty]."""
    b, t_x, t_y = mask.shape
    cum = np.cumsum(duration, 1)
    path = sequence_mask(cum.reshape(b * t_x), t_y).astype(mask.dtype).reshape(b, t_x, t_y)
    shifted = np.pad(path, [[0, 0], [1, 0], [0, 0]])[:, :-1]
    return (path - shifted) * mask


def maximum_path(value, mask, impl="c"):
    """helpers.py:178-194 glue: value*mask, lengths from the mask, native core, int32 path."""
    value = (value * mask).astype(np.float32)
    path = np.zeros_like(value).astype(np.int32)
    t_x = mask.sum(1)[:, 0].astype(np.int32)
    t_y = mask.sum(2)[:, 0].astype(np.int32)
    if impl == "c":
        maximum_path_c(path, value, t_x, t_y)
    elif impl == "ref":
        ref_maximum_path_c()(path, value, t_x, t_y)
    else:
        raise ValueError(impl)
    return path


def maximum_path_numpy(value, mask, max_neg_val=None):
    """helpers.py:197-236 (numpy fallback of the reference), restated."""
    if max_neg_val is None:
        max_neg_val = -np.inf
    value = value * mask
    mask = mask.astype(bool)
    b, t_x, t_y = value.shape
    direction = np.zeros(value.shape, dtype=np.int64)
    v = np.zeros((b, t_x), dtype=np.float32)
    x_range = np.arange(t_x, dtype=np.float32).reshape(1, -1)
    for j in range(t_y):
        v0 = np.pad(v, [[0, 0], [1, 0]], mode="constant", constant_values=max_neg_val)[:, :-1]
        v1 = v
        max_mask = v1 >= v0
        v_max = np.where(max_mask, v1, v0)
        direction[:, :, j] = max_mask
        index_mask = x_range <= j
        v = np.where(index_mask, v_max + value[:, :, j], max_neg_val)
    direction = np.where(mask, direction, 1)
    path = np.zeros(value.shape, dtype=np.float32)
    index = mask[:, :, 0].sum(1).astype(np.int64) - 1
    index_range = np.arange(b)
    for j in reversed(range(t_y)):
        path[index_range, index, j] = 1
        index = index + direction[index_range, index, j] - 1
    return path * mask.astype(np.float32)
