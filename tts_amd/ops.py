"""Thin Python views of the kernel-level C ABI (include/tts_amd.h): argument marshalling only.

Tensors are torch CUDA tensors used purely as device buffers (`data_ptr()`); every FLOP happens in
the HIP kernels of libtts_amd.so.  Weight preparation (weight-norm fold, polyphase/gate/flip
re-ordering, MFMA fragment packing) runs once at load time.
"""
import ctypes

import torch

from . import _lib
from ._lib import P, check, lib, stream_ptr

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH = 0, 1, 1, 2
CONV_NORMAL, CONV_GATE, CONV_SHUFFLE, CONV_COUPLE = 0, 1, 2, 3


class Conv1dArgs(ctypes.Structure):
    """Mirror of `ttsamd_conv1d_args` (include/tts_amd.h)."""

    _fields_ = [
        ("x", ctypes.c_void_p), ("x_bstride", ctypes.c_int64), ("x_rstride", ctypes.c_int64),
        ("c_in", ctypes.c_int32), ("t_in", ctypes.c_int32),
        ("w_packed", ctypes.c_void_p), ("bias", ctypes.c_void_p),
        ("c_out", ctypes.c_int32), ("kernel", ctypes.c_int32), ("dilation", ctypes.c_int32),
        ("pad_left", ctypes.c_int32),
        ("y", ctypes.c_void_p), ("y_bstride", ctypes.c_int64), ("y_rstride", ctypes.c_int64),
        ("t_out", ctypes.c_int32), ("batch", ctypes.c_int32),
        ("in_act", ctypes.c_int32), ("in_slope", ctypes.c_float), ("in_mask", ctypes.c_void_p),
        ("mode", ctypes.c_int32), ("out_act", ctypes.c_int32),
        ("res", ctypes.c_void_p), ("res_bstride", ctypes.c_int64), ("res_rstride", ctypes.c_int64),
        ("accum", ctypes.c_void_p), ("accum_bstride", ctypes.c_int64), ("accum_rstride", ctypes.c_int64),
        ("out_mask", ctypes.c_void_p), ("out_div", ctypes.c_float),
        ("shuffle_u", ctypes.c_int32), ("shuffle_pad", ctypes.c_int32), ("shuffle_t_out", ctypes.c_int32),
    ]


def _dp(t):
    return None if t is None else t.data_ptr()


class PackedConv:
    """A conv layer's weights in MFMA fragment order on the device (+ bias in packed-row order)."""

    def __init__(self, w: torch.Tensor, bias, device, dilation=1, pad_left=None):
        w = w.detach().to("cpu", torch.float32).contiguous()
        self.c_out, self.c_in, self.kernel = w.shape
        self.dilation = dilation
        self.pad_left = (self.kernel - 1) * dilation // 2 if pad_left is None else pad_left
        L = lib()
        if not L.ttsamd_conv1d_supported(self.kernel, dilation):
            raise _lib.TtsAmdError("conv1d kernel=%d dilation=%d has no HIP instantiation" % (self.kernel, dilation))
        L.ttsamd_conv1d_packed_floats.restype = ctypes.c_size_t
        n = L.ttsamd_conv1d_packed_floats(self.c_out, self.c_in, self.kernel)
        packed = torch.empty(n, dtype=torch.float32)
        check(L.ttsamd_conv1d_pack_weights(ctypes.c_void_p(packed.data_ptr()), ctypes.c_void_p(w.data_ptr()),
                                           self.c_out, self.c_in, self.kernel), "conv1d_pack_weights")
        self.w = packed.to(device)
        self.bias = None if bias is None else bias.detach().to(device, torch.float32).contiguous()

    def nbytes(self):
        return self.w.numel() * 4 + (0 if self.bias is None else self.bias.numel() * 4)


def conv1d(pc: PackedConv, x, y, *, t_out=None, c_in_offset=0, in_act=ACT_NONE, in_slope=0.0, in_mask=None,
           mode=CONV_NORMAL, out_act=ACT_NONE, res=None, accum=None, out_mask=None, out_div=0.0,
           y_row_offset=0, res_row_offset=0, accum_row_offset=0, shuffle_u=0, shuffle_pad=0):
    """Launch one fused conv.  x [B, Cx, T_in] (uses channels c_in_offset : c_in_offset + c_in),
    y [B, Cy, T_y] written at rows y_row_offset + packed row (or per `mode`)."""
    B, Cx, T_in = x.shape
    assert x.is_contiguous() and y.is_contiguous() and x.dtype == torch.float32 and y.dtype == torch.float32
    assert c_in_offset + pc.c_in <= Cx
    if t_out is None:
        t_out = T_in + 2 * pc.pad_left - (pc.kernel - 1) * pc.dilation if pc.kernel % 2 == 0 else T_in
    a = Conv1dArgs()
    a.x = x.data_ptr() + 4 * c_in_offset * T_in
    a.x_bstride, a.x_rstride, a.c_in, a.t_in = Cx * T_in, T_in, pc.c_in, T_in
    a.w_packed, a.bias = pc.w.data_ptr(), _dp(pc.bias)
    a.c_out, a.kernel, a.dilation, a.pad_left = pc.c_out, pc.kernel, pc.dilation, pc.pad_left
    Cy, T_y = y.shape[1], y.shape[2]
    a.y = y.data_ptr() + 4 * y_row_offset * T_y
    a.y_bstride, a.y_rstride, a.t_out, a.batch = Cy * T_y, T_y, t_out, B
    a.in_act, a.in_slope, a.in_mask = in_act, in_slope, _dp(in_mask)
    a.mode, a.out_act = mode, out_act
    if res is not None:
        a.res = res.data_ptr() + 4 * res_row_offset * res.shape[2]
        a.res_bstride, a.res_rstride = res.shape[1] * res.shape[2], res.shape[2]
    if accum is not None:
        a.accum = accum.data_ptr() + 4 * accum_row_offset * accum.shape[2]
        a.accum_bstride, a.accum_rstride = accum.shape[1] * accum.shape[2], accum.shape[2]
    a.out_mask, a.out_div = _dp(out_mask), out_div
    a.shuffle_u, a.shuffle_pad, a.shuffle_t_out = shuffle_u, shuffle_pad, T_y
    check(lib().ttsamd_conv1d(ctypes.byref(a), stream_ptr()), "conv1d")
    return y


def fold_weight_norm(sd, name):
    """Effective conv weight from a reference-layout state_dict entry: plain `.weight`, torch>=2.1
    parametrizations (`original0`=g, `original1`=v) or legacy `weight_g/weight_v`.
    w = v * g / ||v|| with the norm over all dims but 0 (torch weight_norm dim=0; for
    ConvTranspose1d dim 0 is in_channels) — SURVEY Appendix B.4.  Load-time glue."""
    if name + ".weight" in sd:
        return sd[name + ".weight"].float()
    g = sd.get(name + ".parametrizations.weight.original0", sd.get(name + ".weight_g"))
    v = sd.get(name + ".parametrizations.weight.original1", sd.get(name + ".weight_v"))
    if g is None or v is None:
        raise KeyError("no weight for %r in state_dict" % name)
    return torch._weight_norm(v.float(), g.float(), 0)


def convt_polyphase_weight(w_t, bias, u):
    """ConvTranspose1d weight [C_in, C_out, 2u] (stride u, pad u/2) -> equivalent 2-tap Conv1d weight
    [C_out*u, C_in, 2] with packed row m = co*u + r:  W'[m, ci, j'] = w[ci, co, r + (1-j')*u]
    (out[co, q*u + r - pad] = sum_ci sum_j x[ci, q-j] w[ci, co, r + j*u])."""
    cin, cout, k = w_t.shape
    assert k == 2 * u, "polyphase path needs kernel == 2*stride (HiFiGAN upsamplers)"
    w = w_t.permute(1, 2, 0).reshape(cout, 2, u, cin)       # [co, j, r, ci]  (k = j*u + r)
    w = w.permute(0, 2, 3, 1)                                # [co, r, ci, j]
    w = torch.flip(w, [3]).reshape(cout * u, cin, 2)         # j' = 1 - j
    b = None if bias is None else bias.repeat_interleave(u)
    return w.contiguous(), b


def gate_permute(w, bias, hidden):
    """Re-order the 2H output rows of a WN in_layer so packed 32-row tile 2a holds tanh channels
    [32a, 32a+32) and tile 2a+1 the matching sigmoid channels (H must be a multiple of 32)."""
    assert hidden % 32 == 0
    idx = []
    for a in range(hidden // 32):
        idx += list(range(32 * a, 32 * a + 32)) + list(range(hidden + 32 * a, hidden + 32 * a + 32))
    idx = torch.tensor(idx)
    return w[idx].contiguous(), (None if bias is None else bias[idx].contiguous())


def replicate_pad(x, y, pad):
    """y = F.pad(x, (pad, pad), 'replicate') on the last dim (hifigan_generator.py:281)."""
    rows = x.numel() // x.shape[-1]
    check(lib().ttsamd_replicate_pad(P(y), P(x), ctypes.c_int64(rows), x.shape[-1], pad, stream_ptr()),
          "replicate_pad")
    return y
