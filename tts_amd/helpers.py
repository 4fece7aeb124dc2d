"""Alignment helpers — host-side mirror of TTS/tts/utils/helpers.py for the hot path.

Same names / argument meaning as the reference (`sequence_mask` helpers.py:43, `generate_path`
helpers.py:154, `maximum_path` helpers.py:172) so the reference's call sites
(`vits.py:919`, `glow_tts.py:241-248,291-297`) can switch by changing the import.  All compute is
HIP (include/tts_amd.h); torch only allocates the output tensors.
"""
import ctypes

import torch

from . import _lib
from ._lib import P, check, lib, stream_ptr


def maximum_path(value: torch.Tensor, mask: torch.Tensor, max_neg_val: float = -1e9) -> torch.Tensor:
    """Monotonic alignment search (helpers.py:172-194 + core.pyx:11-47), entirely on the GPU.

    Shapes: value, mask `[B, T_en, T_de]` -> path `[B, T_en, T_de]` (0/1, dtype of `value`).
    Bit-exact with the reference's Cython core; no D2H/H2D round trip (the reference does one
    each way, helpers.py:187,194).
    """
    _lib.require_gpu(value, "value")
    b, tx, ty = value.shape
    v = value.detach()
    if v.dtype != torch.float32:
        v = v.float()
    v = v.contiguous()
    m = mask.detach().to(torch.float32).contiguous()
    assert m.shape == v.shape
    path = torch.empty((b, tx, ty), dtype=torch.float32, device=v.device)
    if b == 0 or tx == 0 or ty == 0:
        return path.to(value.dtype)
    t_xs = torch.empty(b, dtype=torch.int32, device=v.device)
    t_ys = torch.empty(b, dtype=torch.int32, device=v.device)
    L = lib()
    st = stream_ptr()
    check(L.ttsamd_mask_lengths(P(t_xs), P(t_ys), P(m), b, tx, ty, st), "mask_lengths")
    nbytes = L.ttsamd_maximum_path_workspace_bytes(b, tx, ty)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=v.device)
    flags = 2  # TTSAMD_MAS_PATHS_F32; the kernel writes the zeros too
    check(
        L.ttsamd_maximum_path(P(path), P(v), P(m), P(None), P(t_xs), P(t_ys), b, tx, ty,
                              ctypes.c_float(max_neg_val), P(ws), ctypes.c_size_t(nbytes), flags, st),
        "maximum_path",
    )
    return path if value.dtype == torch.float32 else path.to(value.dtype)


def maximum_path_c(paths: torch.Tensor, values: torch.Tensor, t_xs: torch.Tensor, t_ys: torch.Tensor,
                   max_neg_val: float = -1e9) -> None:
    """Device mirror of core.pyx:42 `maximum_path_c`: in place on `values`, `paths` pre-zeroed int32."""
    _lib.require_gpu(values, "values")
    assert paths.dtype == torch.int32 and values.dtype == torch.float32
    assert paths.is_contiguous() and values.is_contiguous()
    assert t_xs.dtype == torch.int32 and t_ys.dtype == torch.int32
    b, tx, ty = values.shape
    check(
        lib().ttsamd_maximum_path_c(P(paths), P(values), P(t_xs), P(t_ys), b, tx, ty,
                                    ctypes.c_float(max_neg_val), stream_ptr()),
        "maximum_path_c",
    )


def mas_logp(z, m, logs, glow_order=False):
    """logp [B,T_x,T_y] of vits.py:912-918 (glow_tts.py:241-247 with glow_order=True) on the device.
    z [B,C,T_y] latent, m / logs [B,C,T_x] text-side prior statistics."""
    _lib.require_gpu(z, "z")
    z, m, logs = (t.detach().float().contiguous() for t in (z, m, logs))
    B, C, Ty = z.shape
    Tx = m.shape[2]
    assert m.shape == logs.shape == (B, C, Tx)
    logp = torch.empty((B, Tx, Ty), dtype=torch.float32, device=z.device)
    check(lib().ttsamd_mas_logp(P(logp), P(z), P(m), P(logs), B, C, Tx, Ty, int(glow_order), stream_ptr()), "mas_logp")
    return logp


def mas_attention(z, m, logs, x_mask, y_mask, glow_order=False):
    """`attn = maximum_path(logp, attn_mask)` of Vits.forward_mas (vits.py:909-919) / GlowTTS.forward
    (glow_tts.py:236-248), entirely on the GPU.  x_mask [B,1,T_x] or [B,T_x], y_mask likewise -> attn [B,T_x,T_y]."""
    xm = x_mask.reshape(x_mask.shape[0], -1).float()
    ym = y_mask.reshape(y_mask.shape[0], -1).float()
    attn_mask = xm.unsqueeze(-1) * ym.unsqueeze(1)
    return maximum_path(mas_logp(z, m, logs, glow_order), attn_mask)
