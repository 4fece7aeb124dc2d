"""Thin Python views of the kernel-level C ABI (include/tts_amd.h): argument marshalling only.

Tensors are torch CUDA tensors used purely as device buffers (`data_ptr()`); every FLOP happens in
the HIP kernels of libtts_amd.so.  Weight preparation (weight-norm fold, polyphase/gate/flip
re-ordering, MFMA fragment packing) runs once at load time.
"""
import ctypes
import os
import threading
import time

import torch

from . import _lib
from ._lib import P, check, lib, stream_ptr

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH, ACT_GELU = 0, 1, 1, 2, 3
CONV_NORMAL, CONV_GATE, CONV_SHUFFLE, CONV_COUPLE, CONV_RES_SKIP, CONV_COUPLE_AFFINE, CONV_COUPLE_AFFINE_FWD = 0, 1, 2, 3, 4, 5, 6
CONV_COUPLE_AFFINE_MIX = 7       # affine coupling + InvConvNear^-1 + ActNorm^-1 in one epilogue (y2 = the block's mix parameters)


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
        ("y2", ctypes.c_void_p), ("y2_bstride", ctypes.c_int64), ("y2_rstride", ctypes.c_int64),
        ("split_row", ctypes.c_int32), ("row_bias", ctypes.c_void_p), ("w_split", ctypes.c_void_p), ("w_h2", ctypes.c_void_p),
    ]


def _dp(t):
    return None if t is None else t.data_ptr()


class ConvTimer:
    """Optional per-launch HIP-event timing of conv launches (bench.py's live roofline measurement).
    Events are recorded on the stream the kernel is launched on (torch's current stream).  `select(pc, args)`
    picks the launches to time; everything else runs untouched."""

    def __init__(self, select):
        self.select = select
        self.records = []  # (key, flops, bytes, event0, event1)

    def results(self):
        torch.cuda.synchronize()
        out = {}
        for key, flops, nbytes, e0, e1 in self.records:
            r = out.setdefault(key, dict(launches=0, flops=0.0, bytes=0.0, ms=0.0))
            r["launches"] += 1
            r["flops"] += flops
            r["bytes"] += nbytes
            r["ms"] += e0.elapsed_time(e1)
        return out


_TIMER = None


def set_conv_timer(timer):
    global _TIMER
    _TIMER = timer


# Conv arithmetic (TTSAMD_CONV_PRECISION overrides the default):
#   "h2"  (default) large-grid launches split both fp32 operands into two fp16 parts and accumulate three products in fp32
#         (csrc/conv_kernel_h2.h); small-grid launches (single requests) and untuned shapes run the "x3" kernels
#   "x3"  split-bf16 kernels everywhere: three bf16 parts per operand, six products, fp32 accumulate
#   "f32" fp32-input MFMA kernels (bitwise an fmaf chain)
# All three are fp32-class (same parity tolerances; tests/test_conv_gpu.py runs every conv test on each).
PRECISIONS = ("h2", "x3", "f32")
_PRECISION = os.environ.get("TTSAMD_CONV_PRECISION", "h2")
if _PRECISION not in PRECISIONS:
    raise ValueError("TTSAMD_CONV_PRECISION must be one of %s" % (PRECISIONS,))


def set_conv_precision(p):
    global _PRECISION
    if p not in PRECISIONS:
        raise ValueError("conv precision must be one of %s" % (PRECISIONS,))
    _PRECISION = p


def conv_precision():
    return _PRECISION


def set_conv_small_grid(mode):
    """0 .. 3, see ttsamd_conv1d_set_small_grid (include/tts_amd.h); returns the previous mode."""
    return int(lib().ttsamd_conv1d_set_small_grid(int(mode)))


def _pack_h2(w, c_out, c_in, kernel):
    """Host image of ttsamd_conv1d_pack_weights_h2 for a contiguous fp32 [c_out, c_in, kernel] weight."""
    L = lib()
    L.ttsamd_conv1d_packed_h2_bytes.restype = ctypes.c_size_t
    nb = L.ttsamd_conv1d_packed_h2_bytes(c_out, c_in, kernel)
    img = torch.empty(nb, dtype=torch.uint8)
    check(L.ttsamd_conv1d_pack_weights_h2(ctypes.c_void_p(img.data_ptr()), ctypes.c_void_p(w.data_ptr()), c_out, c_in, kernel),
          "conv1d_pack_weights_h2")
    return img


class PackedConv:
    """A conv layer's weights in MFMA fragment order on the device (+ bias in packed-row order): the fp32 image and
    the split-bf16 image."""

    def __init__(self, w: torch.Tensor, bias, device, dilation=1, pad_left=None):
        w = w.detach().to("cpu", torch.float32).contiguous()
        self.c_out, self.c_in, self.kernel = w.shape
        self.dilation = dilation
        self.pad_left = (self.kernel - 1) * dilation // 2 if pad_left is None else pad_left
        L = lib()
        if not L.ttsamd_conv1d_supported(self.kernel, dilation):
            raise _lib.TtsAmdError("conv1d kernel=%d dilation=%d is outside the HIP path's range (kernel <= 31, dilation <= 27)"
                                   % (self.kernel, dilation))
        self.tuned = bool(L.ttsamd_conv1d_tuned(self.kernel, dilation))      # else: the generic split-bf16 kernel
        L.ttsamd_conv1d_packed_floats.restype = ctypes.c_size_t
        n = L.ttsamd_conv1d_packed_floats(self.c_out, self.c_in, self.kernel)
        packed = torch.empty(n, dtype=torch.float32)
        check(L.ttsamd_conv1d_pack_weights(ctypes.c_void_p(packed.data_ptr()), ctypes.c_void_p(w.data_ptr()),
                                           self.c_out, self.c_in, self.kernel), "conv1d_pack_weights")
        self.w = packed.to(device)
        L.ttsamd_conv1d_packed_split_bytes.restype = ctypes.c_size_t
        nb = L.ttsamd_conv1d_packed_split_bytes(self.c_out, self.c_in, self.kernel)
        split = torch.empty(nb, dtype=torch.uint8)
        check(L.ttsamd_conv1d_pack_weights_split(ctypes.c_void_p(split.data_ptr()), ctypes.c_void_p(w.data_ptr()),
                                                 self.c_out, self.c_in, self.kernel), "conv1d_pack_weights_split")
        self.w_split = split.to(device)
        # the two-part fp16 image of the three-product kernels (tuned shapes only: what the large-grid dispatch can reach)
        self.w_h2 = None
        if self.tuned:
            self.w_h2 = _pack_h2(w, self.c_out, self.c_in, self.kernel).to(device)
        self.bias = None if bias is None else bias.detach().to(device, torch.float32).contiguous()
        # square convs of fewer than 32 channels also keep the split image of the weight zero-padded to [32, 32, k]: what the
        # fused ResBlock kernel's 32-channel tile reads (ttsamd_resblock_pair, c = 8 / 16)
        self.w_split_pad32 = self.w_h2_pad32 = None
        if self.c_out == self.c_in and self.c_out in (8, 16):
            wp = torch.zeros(32, 32, self.kernel, dtype=torch.float32)
            wp[: self.c_out, : self.c_in] = w
            nbp = L.ttsamd_conv1d_packed_split_bytes(32, 32, self.kernel)
            sp = torch.empty(nbp, dtype=torch.uint8)
            check(L.ttsamd_conv1d_pack_weights_split(ctypes.c_void_p(sp.data_ptr()), ctypes.c_void_p(wp.data_ptr()), 32, 32,
                                                     self.kernel), "conv1d_pack_weights_split")
            self.w_split_pad32 = sp.to(device)
            self.w_h2_pad32 = _pack_h2(wp, 32, 32, self.kernel).to(device)

    def nbytes(self):
        return self.w.numel() * 4 + (0 if self.bias is None else self.bias.numel() * 4)


def conv1d(pc: PackedConv, x, y, *, t_out=None, c_in_offset=0, in_act=ACT_NONE, in_slope=0.0, in_mask=None,
           mode=CONV_NORMAL, out_act=ACT_NONE, res=None, accum=None, out_mask=None, out_div=0.0,
           y_row_offset=0, res_row_offset=0, accum_row_offset=0, shuffle_u=0, shuffle_pad=0, y2=None, split_row=0,
           row_bias=None):
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
    if y2 is not None and y2.dim() == 3:
        a.y2, a.y2_bstride, a.y2_rstride = y2.data_ptr(), y2.shape[1] * y2.shape[2], y2.shape[2]
    elif y2 is not None:          # CONV_COUPLE_AFFINE_MIX: the flow block's mix parameters (flat)
        a.y2 = y2.data_ptr()
    a.split_row, a.row_bias = split_row, _dp(row_bias)
    # (shapes without a tuned instantiation run on the generic kernel, which is split-bf16 whatever the precision switch says)
    # ... and so does a tuned (k, d) pair in a mode that has no tuned instantiation (a gate conv at k = 3, d = 3 under
    # precision "f32" reached the generic kernel without its image)
    untuned_mode = (mode == CONV_SHUFFLE and pc.kernel != 2) or (mode == CONV_GATE and not (pc.kernel in (3, 5) and pc.dilation == 1)) or \
        (mode not in (CONV_NORMAL, CONV_GATE, CONV_SHUFFLE) and pc.kernel != 1)
    a.w_split = pc.w_split.data_ptr() if (_PRECISION != "f32" or not pc.tuned or untuned_mode) else None
    a.w_h2 = pc.w_h2.data_ptr() if (_PRECISION == "h2" and pc.w_h2 is not None and not untuned_mode) else None
    if _TIMER is not None and not torch.cuda.is_current_stream_capturing():
        key = _TIMER.select(pc, a)
        if key is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib().ttsamd_conv1d(ctypes.byref(a), stream_ptr()), "conv1d")
            e1.record()
            flops = 2.0 * pc.c_out * pc.c_in * pc.kernel * t_out * B
            # algorithmic HBM bytes of this launch: input read once + output written once (+ residual / accumulate
            # operands read once); weights excluded (L2-resident, amortised) — SURVEY.md §8(d)
            n_out = pc.c_out * t_out * B
            nbytes = 4.0 * (pc.c_in * T_in * B + n_out * (1 + (res is not None) + (accum is not None)))
            _TIMER.records.append((key, flops, nbytes, e0, e1))
            return y
    check(lib().ttsamd_conv1d(ctypes.byref(a), stream_ptr()), "conv1d")
    return y


class ResblockArgs(ctypes.Structure):
    """Mirror of `ttsamd_resblock_args` (include/tts_amd.h)."""

    _fields_ = [
        ("x", ctypes.c_void_p), ("y", ctypes.c_void_p), ("accum", ctypes.c_void_p), ("mask", ctypes.c_void_p),
        ("w1_split", ctypes.c_void_p), ("bias1", ctypes.c_void_p), ("w2_split", ctypes.c_void_p), ("bias2", ctypes.c_void_p),
        ("c", ctypes.c_int32), ("t", ctypes.c_int32), ("batch", ctypes.c_int32),
        ("kernel", ctypes.c_int32), ("dilation", ctypes.c_int32),
        ("slope", ctypes.c_float), ("out_div", ctypes.c_float), ("variant", ctypes.c_int32),
        ("w1_bytes", ctypes.c_int64), ("w2_bytes", ctypes.c_int64),
        ("w1_h2", ctypes.c_void_p), ("w2_h2", ctypes.c_void_p), ("w1_h2_bytes", ctypes.c_int64), ("w2_h2_bytes", ctypes.c_int64),
    ]


def resblock_pair_supported(pc1: PackedConv, pc2: PackedConv):
    """True when the fused ResBlock1-iteration kernel covers this conv pair (split-bf16 arithmetic only)."""
    q = lib().ttsamd_resblock_pair_h2_supported if _PRECISION == "h2" else lib().ttsamd_resblock_pair_supported    # h2: also c = 256
    return (_PRECISION in ("x3", "h2") and pc1.c_in == pc1.c_out == pc2.c_in == pc2.c_out and pc1.kernel == pc2.kernel
            and pc2.dilation == 1 and bool(q(pc1.c_out, pc1.kernel, pc1.dilation)))


def resblock_pair(pc1: PackedConv, pc2: PackedConv, x, y, *, slope, mask=None, accum=None, out_div=0.0, variant=0):
    """y = conv2(lrelu(conv1(lrelu(x*mask)) * mask)) + x [+ accum] [/ out_div] as ONE launch (ttsamd_resblock_pair):
    one ResBlock1 iteration, hifigan_generator.py:90-98.  Bitwise equal to the two conv1d launches it replaces (C = 8 / 16 run
    zero-padded on the 32-channel tile: equal up to the sign of zeros)."""
    B, C, T = x.shape
    assert x.is_contiguous() and y.is_contiguous() and y.shape == x.shape and x.dtype == y.dtype == torch.float32
    assert accum is None or (accum.is_contiguous() and accum.shape == x.shape)
    a = ResblockArgs()
    a.x, a.y, a.accum, a.mask = x.data_ptr(), y.data_ptr(), _dp(accum), _dp(mask)
    ws1, ws2 = (pc1.w_split, pc2.w_split) if C >= 32 else (pc1.w_split_pad32, pc2.w_split_pad32)
    a.w1_split, a.bias1, a.w2_split, a.bias2 = ws1.data_ptr(), _dp(pc1.bias), ws2.data_ptr(), _dp(pc2.bias)
    a.w1_bytes, a.w2_bytes = ws1.numel(), ws2.numel()
    if _PRECISION == "h2":
        wh1, wh2 = (pc1.w_h2, pc2.w_h2) if C >= 32 else (pc1.w_h2_pad32, pc2.w_h2_pad32)
        a.w1_h2, a.w2_h2, a.w1_h2_bytes, a.w2_h2_bytes = wh1.data_ptr(), wh2.data_ptr(), wh1.numel(), wh2.numel()
    a.c, a.t, a.batch, a.kernel, a.dilation = C, T, B, pc1.kernel, pc1.dilation
    a.slope, a.out_div, a.variant = slope, out_div, variant
    if _TIMER is not None and not torch.cuda.is_current_stream_capturing():
        sel = getattr(_TIMER, "select_pair", None)
        key = sel(pc1, a) if sel is not None else "fused resblock pair c%d k%d d%d" % (C, pc1.kernel, pc1.dilation)
        if key is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib().ttsamd_resblock_pair(ctypes.byref(a), stream_ptr()), "resblock_pair")
            e1.record()
            flops = 2 * 2.0 * C * C * pc1.kernel * T * B
            # algorithmic HBM bytes of the fused pair: x read once, y written once (+ the accumulate operand)
            nbytes = 4.0 * C * T * B * (2 + (accum is not None))
            _TIMER.records.append((key, flops, nbytes, e0, e1))
            return y
    check(lib().ttsamd_resblock_pair(ctypes.byref(a), stream_ptr()), "resblock_pair")
    return y


_GROUP_SLOT = {3: 0, 7: 1, 11: 2}


def resblock_group_supported(pairs, B, C, T):
    """True when `pairs` [(pc1, pc2), ...] — the MRF branches of one stage at one iteration — can run as ONE launch
    (ttsamd_resblock_group): every pair fusable, kernel sizes distinct and from {3, 7, 11}, one dilation, a small-grid shape."""
    ks = [pc1.kernel for pc1, _ in pairs]
    return (1 < len(pairs) <= 3 and len(set(ks)) == len(ks) and all(k in _GROUP_SLOT for k in ks)
            and len({pc1.dilation for pc1, _ in pairs}) == 1 and all(resblock_pair_supported(pc1, pc2) for pc1, pc2 in pairs)
            and bool(lib().ttsamd_resblock_group_supported(C, T, B)))


def resblock_group(pairs, xs, ys, *, slope, mask=None):
    """ys[i] = ResBlock1 iteration `pairs[i]` of xs[i] (resblock_pair, no accumulate), all branches in one launch."""
    arr = (ResblockArgs * 3)()
    for (pc1, pc2), x, y in zip(pairs, xs, ys):
        B, C, T = x.shape
        assert x.is_contiguous() and y.is_contiguous() and y.shape == x.shape and x.dtype == y.dtype == torch.float32
        a = arr[_GROUP_SLOT[pc1.kernel]]
        a.x, a.y, a.accum, a.mask = x.data_ptr(), y.data_ptr(), None, _dp(mask)
        ws1, ws2 = (pc1.w_split, pc2.w_split) if C >= 32 else (pc1.w_split_pad32, pc2.w_split_pad32)
        a.w1_split, a.bias1, a.w2_split, a.bias2 = ws1.data_ptr(), _dp(pc1.bias), ws2.data_ptr(), _dp(pc2.bias)
        a.w1_bytes, a.w2_bytes = ws1.numel(), ws2.numel()
        a.c, a.t, a.batch, a.kernel, a.dilation = C, T, B, pc1.kernel, pc1.dilation
        a.slope, a.out_div, a.variant = slope, 0.0, 0
    check(lib().ttsamd_resblock_group(arr, stream_ptr()), "resblock_group")
    return ys


def sum_div(srcs, y, div):
    """y = ((srcs[0] + srcs[1]) [+ srcs[2]]) / div — the MRF average over separately written branch outputs."""
    assert 2 <= len(srcs) <= 3 and all(t.is_contiguous() and t.shape == y.shape for t in srcs) and y.is_contiguous()
    check(lib().ttsamd_sum_div(P(y), P(srcs[0]), P(srcs[1]), P(srcs[2]) if len(srcs) > 2 else None, ctypes.c_float(div),
                               ctypes.c_int64(y.numel()), stream_ptr()), "sum_div")
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
    """ConvTranspose1d weight [C_in, C_out, k] (stride u, any k >= 1) -> the equivalent J-tap Conv1d weight [C_out*u, C_in, J],
    J = ceil(k / u), with packed row m = co*u + r (phase r of output channel co):
        out[co, q*u + r - pad] = sum_ci sum_j x[ci, q - j] w[ci, co, r + j*u]      (taps with r + j*u >= k are zero)
    as a Conv1d with pad_left = J - 1:  W'[m, ci, j'] = w[ci, co, r + (J - 1 - j')*u].  HiFiGAN's k = 2u gives J = 2."""
    cin, cout, k = w_t.shape
    J = -(-k // u)
    wz = torch.zeros(cin, cout, J * u, dtype=w_t.dtype)
    wz[:, :, :k] = w_t
    w = wz.permute(1, 2, 0).reshape(cout, J, u, cin)         # [co, j, r, ci]  (tap index = j*u + r)
    w = w.permute(0, 2, 3, 1)                                # [co, r, ci, j]
    w = torch.flip(w, [3]).reshape(cout * u, cin, J)         # j' = J - 1 - j
    b = None if bias is None else bias.repeat_interleave(u)
    return w.contiguous(), b


PAIR_ROWS = 16     # paired-row conv modes (GATE, COUPLE_AFFINE*): a packed 32-row tile = 16 first halves + the 16 matching second halves


def pair_index(n, second_offset):
    """Packed-row order of the paired modes for n output channels: tile m holds rows [16m, 16m+16) of the first operand
    (tanh / t) followed by the same channels of the second (sigmoid / s, at +second_offset in the source); -1 marks the zero
    rows that pad a last partial tile."""
    idx = []
    for a in range((n + PAIR_ROWS - 1) // PAIR_ROWS):
        lo, cnt = PAIR_ROWS * a, min(PAIR_ROWS, n - PAIR_ROWS * a)
        idx += list(range(lo, lo + cnt)) + [-1] * (PAIR_ROWS - cnt)
        idx += list(range(second_offset + lo, second_offset + lo + cnt)) + [-1] * (PAIR_ROWS - cnt)
    return idx


def pair_permute(w, bias, n, second_offset):
    """Rows of a [2n(+), c_in, k] weight (first operand rows [0, n), second at [second_offset, second_offset + n)) -> the
    paired modes' packed order (zero rows where a last tile is partial)."""
    idx = torch.tensor(pair_index(n, second_offset))
    keep = idx >= 0
    wp = torch.zeros((idx.numel(),) + tuple(w.shape[1:]), dtype=w.dtype)
    wp[keep] = w[idx[keep]]
    bp = None
    if bias is not None:
        bp = torch.zeros(idx.numel(), dtype=bias.dtype)
        bp[keep] = bias[idx[keep]]
    return wp.contiguous(), bp


def gate_permute(w, bias, hidden):
    """Re-order the 2H output rows of a WN in_layer into the GATE mode's packed order: 32-row tile m = tanh channels
    [16m, 16m+16) then the matching sigmoid channels (include/tts_amd.h)."""
    return pair_permute(w, bias, hidden, hidden)


class CopySeg(ctypes.Structure):
    """Mirror of `ttsamd_copy_seg` (include/tts_amd.h)."""

    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("d0", ctypes.c_int32), ("d1", ctypes.c_int32),
                ("d2", ctypes.c_int32), ("s0", ctypes.c_int64), ("s1", ctypes.c_int64), ("s2", ctypes.c_int64),
                ("t0", ctypes.c_int64), ("t1", ctypes.c_int64), ("t2", ctypes.c_int64), ("elem_bytes", ctypes.c_int32)]


COPY_MAX_SEGS = 12


def _copy_seg(d, v):
    shp = [1] * (3 - v.dim()) + list(v.shape)
    g = CopySeg()
    g.src, g.dst = v.data_ptr(), d.data_ptr()
    g.d0, g.d1, g.d2 = shp
    g.s0, g.s1, g.s2 = [0] * (3 - v.dim()) + list(v.stride())
    g.t0, g.t1, g.t2 = [0] * (3 - d.dim()) + list(d.stride())
    g.elem_bytes = d.element_size()
    return g


def _copy_launch(segs):
    for i in range(0, len(segs), COPY_MAX_SEGS):
        part = segs[i:i + COPY_MAX_SEGS]
        arr = (CopySeg * len(part))(*part)
        check(lib().ttsamd_copy_strided(arr, len(part), stream_ptr()), "copy_strided")


def clone_views(views):
    """[v.clone(memory_format=contiguous) for v in views] as ONE launch per 12 views (ttsamd_copy_strided): the slices /
    transposes of a graph's static buffers a request hands out.  Views of up to 3 dims, 4- or 8-byte elements; None passes
    through."""
    outs, segs = [], []
    for v in views:
        if v is None:
            outs.append(None)
            continue
        assert v.dim() <= 3 and v.element_size() in (4, 8), "clone_views: <= 3-D views of 4- / 8-byte elements"
        o = torch.empty(v.shape, dtype=v.dtype, device=v.device)
        outs.append(o)
        if o.numel():
            segs.append(_copy_seg(o, v))
    _copy_launch(segs)
    return outs


def copy_into(dsts, srcs):
    """dst.copy_(src) for every pair as ONE launch per 12 pairs (same shape and dtype, up to 3 dims, any strides)."""
    segs = []
    for d, v in zip(dsts, srcs):
        assert d.shape == v.shape and d.dtype == v.dtype and v.dim() <= 3 and d.element_size() in (4, 8)
        if d.numel():
            segs.append(_copy_seg(d, v))
    _copy_launch(segs)


MASK_MAX_STAGES = 8       # TTSAMD_MASK_MAX_STAGES (include/tts_amd.h)


def stage_masks(lengths, scales, t_stage, quantum=1, add=0):
    """One launch for every length mask of a ragged vocoder call -> ([mask_s float [B, t_stage[s]]], len_eff int64 [B]);
    len_eff = lengths // quantum * quantum + add, mask_s[b, t] = t < len_eff[b] * scales[s]   (ttsamd_stage_masks)."""
    lengths = lengths.to(torch.int64).contiguous()
    B, n = lengths.shape[0], len(scales)
    flat = torch.empty(B * sum(int(t) for t in t_stage), dtype=torch.float32, device=lengths.device)
    len_eff = torch.empty(B, dtype=torch.int64, device=lengths.device)
    # the kernel takes up to MASK_MAX_STAGES stages per launch (a generator with more than 7 upsample layers — the reference
    # accepts any list, hifigan_generator.py:199-233 — needs more): chunks of 8 stages, each recomputing the same len_eff
    off = 0
    out = []
    for s0 in range(0, max(n, 1), MASK_MAX_STAGES):
        sc_l, ts_l = [int(v) for v in scales[s0:s0 + MASK_MAX_STAGES]], [int(v) for v in t_stage[s0:s0 + MASK_MAX_STAGES]]
        m = len(sc_l)
        sc, ts = (ctypes.c_int32 * max(m, 1))(*sc_l), (ctypes.c_int32 * max(m, 1))(*ts_l)
        part = flat[off:]
        check(lib().ttsamd_stage_masks(P(part), P(len_eff), P(lengths), B, int(quantum), int(add), sc, ts, m, stream_ptr()),
              "stage_masks")
        for t in ts_l:
            out.append(flat[off: off + B * t].view(B, t))
            off += B * t
    return out, len_eff


def replicate_pad(x, y, pad, lengths=None, len_bias=0):
    """y = F.pad(x, (pad, pad), 'replicate') on the last dim (hifigan_generator.py:281); with `lengths` [B] (int64)
    every item of the ragged batch replicates its own last valid frame."""
    if lengths is not None:
        B, C, T = x.shape
        check(lib().ttsamd_replicate_pad_ragged_ex(P(y), P(x), P(lengths.to(torch.int64).contiguous()),
                                                   ctypes.c_int64(int(len_bias)), B, C, T, pad, stream_ptr()), "replicate_pad_ragged")
        return y
    rows = x.numel() // x.shape[-1]
    check(lib().ttsamd_replicate_pad(P(y), P(x), ctypes.c_int64(rows), x.shape[-1], pad, stream_ptr()),
          "replicate_pad")
    return y


class NormArgs(ctypes.Structure):
    """Mirror of `ttsamd_norm_args` (include/tts_amd.h)."""

    _fields_ = [
        ("x", ctypes.c_void_p), ("x_bstride", ctypes.c_int64), ("x_rstride", ctypes.c_int64),
        ("c", ctypes.c_int32), ("t", ctypes.c_int32), ("batch", ctypes.c_int32),
        ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("eps", ctypes.c_float),
        ("dw_w", ctypes.c_void_p), ("dw_bias", ctypes.c_void_p),
        ("dw_kernel", ctypes.c_int32), ("dw_dilation", ctypes.c_int32),
        ("in_mask", ctypes.c_void_p),
        ("pre_res", ctypes.c_void_p), ("pre_bstride", ctypes.c_int64), ("pre_rstride", ctypes.c_int64),
        ("act", ctypes.c_int32),
        ("post_res", ctypes.c_void_p), ("post_bstride", ctypes.c_int64), ("post_rstride", ctypes.c_int64),
        ("out_mask", ctypes.c_void_p),
        ("y", ctypes.c_void_p), ("y_bstride", ctypes.c_int64), ("y_rstride", ctypes.c_int64),
    ]


def channel_norm(x, y, gamma, beta, eps, *, dw_w=None, dw_bias=None, dw_dilation=1, in_mask=None, pre_res=None,
                 act=ACT_NONE, post_res=None, out_mask=None):
    """y = [post_res +] act(LN_c([dwconv](x) [+ pre_res])) [* out_mask] on [B, C, T] (include/tts_amd.h)."""
    B, C, T = x.shape
    assert x.is_contiguous() and y.is_contiguous() and y.shape == x.shape
    a = NormArgs()
    a.x, a.x_bstride, a.x_rstride, a.c, a.t, a.batch = x.data_ptr(), C * T, T, C, T, B
    a.gamma, a.beta, a.eps = gamma.data_ptr(), beta.data_ptr(), eps
    if dw_w is not None:
        a.dw_w, a.dw_bias, a.dw_kernel, a.dw_dilation = dw_w.data_ptr(), _dp(dw_bias), dw_w.shape[-1], dw_dilation
    a.in_mask = _dp(in_mask)
    if pre_res is not None:
        assert pre_res.shape == x.shape and pre_res.is_contiguous()
        a.pre_res, a.pre_bstride, a.pre_rstride = pre_res.data_ptr(), C * T, T
    a.act = act
    if post_res is not None:
        assert post_res.shape == x.shape and post_res.is_contiguous()
        a.post_res, a.post_bstride, a.post_rstride = post_res.data_ptr(), C * T, T
    a.out_mask = _dp(out_mask)
    a.y, a.y_bstride, a.y_rstride = y.data_ptr(), C * T, T
    check(lib().ttsamd_channel_norm(ctypes.byref(a), stream_ptr()), "channel_norm")
    return y


def rel_attention(qkv, out, mask, heads, emb_rel_k=None, emb_rel_v=None, window=0):
    """qkv [B, 3*H*dk, T] (rows: q | k | v) -> out [B, H*dk, T] (include/tts_amd.h: ttsamd_rel_attention)."""
    B, C3, T = qkv.shape
    C = C3 // 3
    dk = C // heads
    assert qkv.is_contiguous() and out.is_contiguous() and out.shape == (B, C, T)
    base = qkv.data_ptr()
    check(lib().ttsamd_rel_attention(P(out), ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * C * T),
                                     ctypes.c_void_p(base + 8 * C * T), ctypes.c_int64(C3 * T), P(mask),
                                     P(emb_rel_k), P(emb_rel_v), window, B, heads, dk, T, stream_ptr()),
          "rel_attention")
    return out


def embed(tokens, emb, mask, scale, y):
    B, T = tokens.shape
    V, C = emb.shape
    assert tokens.dtype == torch.int64 and tokens.is_contiguous() and y.shape == (B, C, T)
    check(lib().ttsamd_embed(P(y), P(tokens), P(emb), P(mask), ctypes.c_float(scale), B, C, T, V, stream_ptr()), "embed")
    return y


def embed_cat(tokens, emb, mask, scale, extra, y):
    """y [B, C+E, T]: embedding rows then extra[b] ([B,E], e.g. the language embedding) broadcast over time."""
    B, T = tokens.shape
    V, C = emb.shape
    E = extra.shape[1]
    assert tokens.dtype == torch.int64 and tokens.is_contiguous() and y.shape == (B, C + E, T)
    check(lib().ttsamd_embed_cat(P(y), P(tokens), P(emb), P(mask), ctypes.c_float(scale), P(extra.contiguous().float()), E,
                                 B, C, T, V, stream_ptr()), "embed_cat")
    return y


def sequence_mask(lengths, t, out=None):
    """helpers.py:43-57 on the device; returns float [B, t] (written into `out` when given)."""
    lengths = lengths.to(torch.int64).contiguous()
    mask = torch.empty((lengths.shape[0], t), dtype=torch.float32, device=lengths.device) if out is None else out
    check(lib().ttsamd_sequence_mask(P(mask), P(lengths), lengths.shape[0], t, stream_ptr()), "sequence_mask")
    return mask


def convflow_pre(h, z, z_ch, w, bias, g):
    B, C, T = h.shape
    check(lib().ttsamd_convflow_pre(P(h), P(z), z_ch, P(w), P(bias), P(g), B, C, T, stream_ptr()), "convflow_pre")
    return h


def convflow_spline_reverse(z_out, z_in, h, mask, num_bins, filter_channels, tail_bound):
    B, _, T = z_in.shape
    assert h.shape == (B, 3 * num_bins - 1, T) and h.is_contiguous()
    check(lib().ttsamd_convflow_spline_reverse(P(z_out), P(z_in), P(h), P(mask), B, T, num_bins,
                                               ctypes.c_float(filter_channels), ctypes.c_float(tail_bound),
                                               stream_ptr()), "convflow_spline_reverse")
    return z_out


def sdp_affine_reverse(z_out, z_in, m, logs, mask):
    B, _, T = z_in.shape
    check(lib().ttsamd_sdp_affine_reverse(P(z_out), P(z_in), P(m), P(logs), P(mask), B, T, stream_ptr()),
          "sdp_affine_reverse")
    return z_out


class _HostLengths:
    """Pool of pinned (host-mapped) int64 mirrors the durations kernel writes y_lengths into: the host polls the mirror
    instead of `y_lengths.max().item()` (a reduce kernel + device-to-host copy + a blocking stream synchronise)."""

    _tls = threading.local()
    quarantine = []          # mirrors a stalled kernel may still write (durations() timed out): referenced for the process lifetime

    @classmethod
    def take(cls, n):
        pool = cls._tls.__dict__.setdefault("pool", {})
        free = pool.setdefault(n, [])
        if free:
            return free.pop()
        t = torch.empty(n, dtype=torch.int64, pin_memory=True)
        return t, t.numpy()

    @classmethod
    def give(cls, n, item):
        cls._tls.pool[n].append(item)


def durations(logw, mask, length_scale, glow=False, durations_in=None, t_valid=None, want_max=False, out=None, host_out=None):
    """-> (w_ceil float [B,T], cum int32 [B,T], y_lengths int64 [B]) (include/tts_amd.h: ttsamd_durations_ex).
    t_valid: columns >= t_valid own no frames (text-length bucket padding).  want_max=True also returns max(y_lengths) as a
    Python int, read from a pinned host mirror the kernel writes (the request's ONE host wait) — not under stream capture."""
    src = logw if logw is not None else durations_in
    B, T = src.shape[0], src.shape[-1]
    dev = src.device
    if out is not None:         # (dur float [B,T], cum int32 [B,T], y_lengths int64 [B]): a model's per-stream scratch
        dur, cum, ylen = out
    else:
        dur = torch.empty((B, T), dtype=torch.float32, device=dev)
        cum = torch.empty((B, T), dtype=torch.int32, device=dev)
        ylen = torch.empty((B,), dtype=torch.int64, device=dev)
    host = None
    if want_max:
        if torch.cuda.is_current_stream_capturing():      # the host wait below would poll for ever: nothing runs during a capture
            raise _lib.TtsAmdError("durations(want_max=True) waits for the device: not valid under stream capture")
        host = _HostLengths.take(B)
        host[1][:] = -1
    # glow: 0 VITS, 1 Glow, 2 Glow ragged-exact
    check(lib().ttsamd_durations_ex(P(dur), P(cum), P(ylen), P(host[0]) if host else None, P(logw), P(durations_in), P(mask),
                                    ctypes.c_float(length_scale), int(glow), T if t_valid is None else int(t_valid), B, T,
                                    stream_ptr()), "durations")
    if not want_max:
        return dur, cum, ylen
    arr = host[1]
    deadline = None
    spins = 0
    while int(arr.min()) < 0:
        spins += 1
        if spins & 0x3F == 0:
            time.sleep(0)                   # give the GIL away: other request lanes' host threads issue their launches meanwhile
        if spins & 0xFFF == 0:              # a stalled stream must not hang the host for ever: ~ every millisecond look at the clock
            now = time.monotonic()
            if deadline is None:
                deadline = now + 30.0
            elif now > deadline:
                # the stalled kernel may still write the mirror later: keep it referenced (never back to the pool, never freed
                # to the pinned-memory allocator) instead of letting another tensor take its place
                _HostLengths.quarantine.append(host)
                raise _lib.TtsAmdError("durations: the device did not publish y_lengths within 30 s (stalled stream?)")
    t_max = int(arr.max())
    if host_out is not None:
        host_out["y_lengths"] = [int(v) for v in arr]        # every item's frame count, already on the host
    _HostLengths.give(B, host)
    return dur, cum, ylen, t_max


def generate_path(cum, x_mask, y_lengths, t_y):
    """helpers.py:154-169 from cumulative durations; returns float [B, T_x, t_y]."""
    B, Tx = cum.shape
    attn = torch.empty((B, Tx, t_y), dtype=torch.float32, device=cum.device)
    check(lib().ttsamd_generate_path(P(attn), P(cum), P(x_mask), P(y_lengths), B, Tx, t_y, stream_ptr()), "generate_path")
    return attn


def expand_prior(m, logs, noise, cum, x_mask, y_lengths, t_y, noise_scale, mask_out=False, want_stats=True,
                 second_copy=False, noise_packed=False):
    """vits.py:1152-1155 / glow_tts.py:137-148,361 as one gather -> dict(z_p, z_p2, m_p, logs_p, y_mask).
    m / logs [B,C,T_x] may be channel-slices of one projection buffer (row stride T_x, any batch stride)."""
    B, C, Tx = m.shape
    dev = m.device
    assert m.stride(2) == 1 and m.stride(1) == Tx and (logs is None or logs.stride() == m.stride())
    new = lambda: torch.empty((B, C, t_y), dtype=torch.float32, device=dev)  # noqa: E731
    z_p = new()
    z_p2 = new() if second_copy else None
    m_p = new() if want_stats else None
    logs_p = new() if want_stats else None
    y_mask = torch.empty((B, t_y), dtype=torch.float32, device=dev)
    # noise_packed: `noise` holds a contiguous [B, C, max(y_lengths)] draw at the head of a larger buffer (the draw at the
    # reference's shape, placed in a graph's fixed scratch); columns beyond that extent read as zero
    check(lib().ttsamd_expand_prior_ex(P(z_p), P(z_p2), P(m_p), P(logs_p), P(y_mask), P(m), P(logs), ctypes.c_int64(m.stride(0)), P(noise), P(cum),
                                       P(x_mask), P(y_lengths), ctypes.c_float(noise_scale), int(mask_out), int(noise_packed), B, C, Tx,
                                       t_y, stream_ptr()), "expand_prior")
    return {"z_p": z_p, "z_p2": z_p2, "m_p": m_p, "logs_p": logs_p, "y_mask": y_mask}


def scale(x, s):
    y = torch.empty_like(x)
    check(lib().ttsamd_scale(P(y), P(x.contiguous()), ctypes.c_float(s), ctypes.c_int64(x.numel()), stream_ptr()), "scale")
    return y


def glow_squeeze(x, mask, n):
    """decoder.py:8-28 -> (x_sqz [B,C*n,T//n], mask_sqz [B,T//n])."""
    B, C, T = x.shape
    Tq = T // n
    y = torch.empty((B, C * n, Tq), dtype=torch.float32, device=x.device)
    mq = torch.empty((B, Tq), dtype=torch.float32, device=x.device)
    check(lib().ttsamd_glow_squeeze(P(y), P(mq), P(x), P(mask), B, C, T, n, stream_ptr()), "glow_squeeze")
    return y, mq


def glow_unsqueeze(x, mask_q, n, t_out):
    """decoder.py:31-47 -> [B, Cq//n, t_out] (t_out >= Tq*n; an odd dropped frame comes back as zeros)."""
    B, Cq, Tq = x.shape
    y = torch.empty((B, Cq // n, t_out), dtype=torch.float32, device=x.device)
    check(lib().ttsamd_glow_unsqueeze(P(y), P(x), P(mask_q), B, Cq, Tq, n, t_out, stream_ptr()), "glow_unsqueeze")
    return y


def glow_invconv_actnorm(x, w_inv, bias, logs, mask, num_splits=4, forward=False):
    """reverse: invconv(w_inv) then actnorm^-1; forward: actnorm then invconv(w) (pass the weight itself as w_inv)."""
    B, C, T = x.shape
    check(lib().ttsamd_glow_invconv_actnorm(P(x), P(w_inv), P(bias), P(logs), P(mask), B, C, T, num_splits, int(forward),
                                            stream_ptr()), "glow_invconv_actnorm")
    return x


def row_sum(x):
    """sum over the last dim of [..., T] -> [...]  (attn.sum(-1))."""
    x = x.float().contiguous()
    o = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    check(lib().ttsamd_row_sum(P(o), P(x), ctypes.c_int64(o.numel()), x.shape[-1], stream_ptr()), "row_sum")
    return o


def attn_durations(cum, x_mask, y_lengths):
    B, Tx = cum.shape
    o = torch.empty((B, Tx), dtype=torch.float32, device=cum.device)
    check(lib().ttsamd_attn_durations(P(o), P(cum), P(x_mask), P(y_lengths), B, Tx, stream_ptr()), "attn_durations")
    return o


def l2_normalize(x, eps=1e-12):
    """F.normalize(x) over the last dim of a [rows, cols] tensor (vits.py:882)."""
    x = x.float().contiguous()
    y = torch.empty_like(x)
    check(lib().ttsamd_l2_normalize(P(y), P(x), x.shape[0], x.shape[1], ctypes.c_float(eps), stream_ptr()), "l2_normalize")
    return y


def add_row_bias(x, rb):
    """x [B,C,T] + rb [B,C] broadcast over time."""
    B, C, T = x.shape
    y = torch.empty_like(x)
    check(lib().ttsamd_add_row_bias(P(y), P(x), P(rb.contiguous()), ctypes.c_int64(B * C), T, stream_ptr()), "add_row_bias")
    return y


def speaker_cond(pc: PackedConv, g):
    """1x1 conv of the speaker vector g [B,Cg,1] -> per-(b, row) offsets [B, c_out] (a `row_bias` operand)."""
    B = g.shape[0]
    out = torch.empty((B, pc.c_out, 1), dtype=torch.float32, device=g.device)
    conv1d(pc, g.contiguous().float(), out)
    return out.reshape(B, pc.c_out)


def linear_interp(x, scale_factor, recompute_scale_factor=False):
    """F.interpolate(x, scale_factor=[s], mode="linear") on [B, C, T] (hifigan_decoder.py:688-700, vits.py:952).
    recompute_scale_factor=True maps coordinates with out_size/in_size instead of s (torch semantics; used by
    interpolate_vocoder_input, vocoder/utils/generic_utils.py:24-26)."""
    import math

    B, C, T = x.shape
    t_out = int(math.floor(T * float(scale_factor)))
    if recompute_scale_factor and T > 0:
        scale_factor = t_out / T
    x = x.float().contiguous()
    y = torch.empty((B, C, t_out), dtype=torch.float32, device=x.device)
    check(lib().ttsamd_linear_interp(P(y), P(x), ctypes.c_int64(B * C), T, t_out, ctypes.c_double(float(scale_factor)),
                                     stream_ptr()), "linear_interp")
    return y


def sample_gaussian(stats, noise, mask):
    """z = (mean + noise * exp(log_scale)) * mask with stats [B,2C,T] = mean | log_scale (networks.py:286-287)."""
    B, C2, T = stats.shape
    z = torch.empty((B, C2 // 2, T), dtype=torch.float32, device=stats.device)
    check(lib().ttsamd_sample_gaussian(P(z), P(stats), P(noise.contiguous()), P(mask), B, C2 // 2, T, stream_ptr()),
          "sample_gaussian")
    return z
