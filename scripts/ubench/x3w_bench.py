"""Driver of the products-per-output experiment (scripts/ubench/x3w.hip): accuracy against an fp64 conv (random and the
adversarial operands of tests/test_conv_gpu.py) and time against the library's direct split-bf16 kernel, same box, same tensors.
    (cd scripts/ubench && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off x3w.hip -o libx3w.so)
    python scripts/ubench/x3w_bench.py"""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from tts_amd import _lib, ops  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
X3W = ctypes.CDLL(os.path.join(HERE, "libx3w.so"))
dev = torch.device("cuda:0")


def pack_transformed(w):
    """w [Co, Ci, K] fp32 -> split image of U [Co, Ci, 4 G] (fp64 transform, one rounding to fp32)."""
    Co, Ci, K = w.shape
    G = (K + 2) // 3
    wd = torch.zeros(Co, Ci, 3 * G, dtype=torch.float64)
    wd[:, :, :K] = w.double()
    U = torch.empty(Co, Ci, 4 * G, dtype=torch.float64)
    for g in range(G):
        g0, g1, g2 = wd[:, :, 3 * g], wd[:, :, 3 * g + 1], wd[:, :, 3 * g + 2]
        U[:, :, 4 * g + 0] = g0
        U[:, :, 4 * g + 1] = (g0 + g1 + g2) / 2
        U[:, :, 4 * g + 2] = (g0 - g1 + g2) / 2
        U[:, :, 4 * g + 3] = g2
    U = U.float().contiguous()
    L = _lib.lib()
    L.ttsamd_conv1d_packed_split_bytes.restype = ctypes.c_size_t
    nb = L.ttsamd_conv1d_packed_split_bytes(Co, Ci, 4 * G)
    img = torch.empty(nb, dtype=torch.uint8)
    _lib.check(L.ttsamd_conv1d_pack_weights_split(ctypes.c_void_p(img.data_ptr()), ctypes.c_void_p(U.data_ptr()), Co, Ci, 4 * G), "pack")
    return img.to(dev)


def x3w(x, img, bias, res, K, slope=1.0):
    B, C, T = x.shape
    y = torch.empty_like(x)
    rc = X3W.x3w_conv(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(img.data_ptr()), ctypes.c_void_p(bias.data_ptr() if bias is not None else 0),
                      ctypes.c_void_p(res.data_ptr() if res is not None else 0), ctypes.c_void_p(y.data_ptr()), C, T, B, K,
                      ctypes.c_float(slope), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    return y


def time_us(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def f32_from_bits(sign, exp, mant):
    bits = (sign.astype(np.uint32) << 31) | (exp.astype(np.uint32) << 23) | mant.astype(np.uint32)
    return torch.from_numpy(bits.view(np.float32).copy())


def max_residual_values(rng, shape, e_lo, e_hi):
    n = int(np.prod(shape))
    mant = (rng.integers(0, 128, n) << 16) | (rng.choice([0x7F, 0x80, 0x7E, 0x81], n) << 8) | rng.choice([0x7F, 0x80, 0xFF, 0x01], n)
    return f32_from_bits(rng.integers(0, 2, n), rng.integers(e_lo, e_hi + 1, n), mant).reshape(shape)


def conv64(x, w, K):
    return F.conv1d(x.double(), w.double(), None, padding=(K - 1) // 2)


print("== accuracy: max |err| / sum|w x| against an fp64 conv (2^-20 = %.2e) ==" % 2.0 ** -20)
V2 = bool(os.environ.get("X3W_V2"))
print("kernel version:", "2 (8 waves, double-buffered, pipelined transform)" if V2 else "1 (4 waves, single buffer)")
for name, C, K, T in (("max-residual operands", 128, 11, 260), ("max-residual operands", 128, 7, 400), ("max-residual operands", 128, 3, 300),
                      ("randn", 256, 11, 300), ("alternating-sign cancellation", 256, 11, 300)):
    if V2 and K == 3:
        continue
    rng = np.random.default_rng(C + K + T)
    if name.startswith("max"):
        x = max_residual_values(rng, (1, C, T), 120, 130)
        w = max_residual_values(rng, (C, C, K), 115, 122)
    elif name == "randn":
        x = torch.randn(1, C, T, generator=torch.Generator().manual_seed(1))
        w = torch.randn(C, C, K, generator=torch.Generator().manual_seed(2)) / np.sqrt(C * K)
    else:
        sign = np.where((np.arange(C)[:, None] + np.arange(T)[None, :]) % 2 == 0, 1.0, -1.0)
        x = torch.from_numpy((sign * (1.0 + 1e-3 * rng.standard_normal((C, T)))).astype(np.float32))[None]
        w = torch.from_numpy((0.05 * (1.0 + 1e-3 * rng.standard_normal((C, C, K)))).astype(np.float32))
    want = conv64(x, w, K)
    scale = conv64(x.abs(), w.abs(), K)
    yw = x3w(x.to(dev), pack_transformed(w), None, None, K).cpu().double()
    yd = torch.empty(1, C, T, device=dev)
    ops.conv1d(ops.PackedConv(w, None, dev), x.to(dev), yd)
    ew = float(((yw - want).abs() / scale).max())
    ed = float(((yd.cpu().double() - want).abs() / scale).max())
    rel = float((yw - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
    print("%-34s C=%3d k=%2d: F(2,3) %.3e   direct (library) %.3e   ratio %.2f   rel RMS of F(2,3) %.2e" % (name, C, K, ew, ed, ew / ed, rel))

print("== time at the headline shapes (B = 32; bias + residual as in a ResBlock conv) ==")
for C, K, T in ((128, 11, 49280), (256, 11, 6160), (128, 7, 49280), (256, 7, 6160), (256, 3, 6160), (128, 3, 49280)):
    if V2 and K == 3:
        continue
    B = 32
    g = torch.Generator().manual_seed(C + K)
    x = torch.randn(B, C, T, generator=g).to(dev)
    res = torch.randn(B, C, T, generator=g).to(dev)
    w = torch.randn(C, C, K, generator=g) / np.sqrt(C * K)
    bias = torch.randn(C, generator=g)
    img = pack_transformed(w)
    pc = ops.PackedConv(w, bias, dev)
    y = torch.empty_like(x)
    bd = bias.to(dev)
    t_w = time_us(lambda: x3w(x, img, bd, res, K, 0.1))
    t_d = time_us(lambda: ops.conv1d(pc, x, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=res))
    yw = x3w(x, img, bd, res, K, 0.1)
    ops.conv1d(pc, x, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=res)
    rel = float((yw - y).double().pow(2).mean().sqrt() / y.double().pow(2).mean().sqrt())
    fl = 2.0 * C * C * K * T * B
    print("C=%3d k=%2d T=%5d: direct %7.1f us (%5.1f TF-eq)   F(2,3) %7.1f us (%5.1f TF-eq)   speed-up %.3f   rel diff %.1e"
          % (C, K, T, t_d, fl / t_d / 1e6, t_w, fl / t_w / 1e6, t_d / t_w, rel))

print("== the same launches on all-zero operands (no switching activity: the socket stays below its power limit, full clock) ==")
for C, K, T in ((128, 11, 49280), (128, 7, 49280)):
    B = 32
    x = torch.zeros(B, C, T, device=dev)
    res = torch.zeros(B, C, T, device=dev)
    w = torch.zeros(C, C, K)
    img = pack_transformed(w)
    pc = ops.PackedConv(w, torch.zeros(C), dev)
    y = torch.empty_like(x)
    bd = torch.zeros(C, device=dev)
    t_w = time_us(lambda: x3w(x, img, bd, res, K, 0.1))
    t_d = time_us(lambda: ops.conv1d(pc, x, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=res))
    print("C=%3d k=%2d T=%5d zeros: direct %7.1f us   F(2,3) %7.1f us   speed-up %.3f" % (C, K, T, t_d, t_w, t_d / t_w))
