"""Driver of the three-product experiment (scripts/ubench/h2.hip): accuracy against an fp64 conv (random and adversarial operands)
and time against the library's direct six-product split-bf16 kernel, same box, same tensors.
    (cd scripts/ubench && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off h2.hip -o libh2.so)
    python scripts/ubench/h2_bench.py"""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from tts_amd import ops  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
H2 = ctypes.CDLL(os.path.join(HERE, "libh2.so"))
dev = torch.device("cuda:0")


def pack_h2(w):
    """w [Co, Ci, K] fp32 -> (image [m-tile][chunk][tap][part][64 lanes][8 fp16] + slack, row_scale [Co]).
    Rows are scaled by a power of two so that the row maximum lies in [2^13, 2^14)."""
    w = w.numpy().astype(np.float32)
    Co, Ci, K = w.shape
    mx = np.abs(w).reshape(Co, -1).max(1)
    e = np.where(mx > 0, 13 - np.floor(np.log2(np.maximum(mx, 1e-45))), 0.0)
    rs = np.exp2(e).astype(np.float32)
    ws = (w * rs[:, None, None]).astype(np.float32)
    assert np.isfinite(ws).all()
    hi = ws.astype(np.float16)
    lo = ((ws - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    mt, nch = Co // 32, Ci // 16
    img = np.zeros((mt * nch * K + 2, 2, 64, 8), np.float16)
    for part, p in enumerate((hi, lo)):
        a = p.reshape(mt, 32, nch, 2, 8, K)                 # (mt, r, c, h, i, tap)
        a = a.transpose(0, 2, 5, 3, 1, 4)                   # (mt, c, tap, h, r, i)
        img[:mt * nch * K, part] = a.reshape(mt * nch * K, 64, 8)
    return torch.from_numpy(img.view(np.uint8).reshape(-1)).to(dev), torch.from_numpy(rs)


def h2(x, img, rs, bias, res, K, slope=1.0, x_scale=1.0):
    B, C, T = x.shape
    y = torch.empty_like(x)
    ru = (1.0 / (rs.double() * x_scale)).float().to(dev)
    rsd = rs.to(dev)
    rc = H2.h2_conv(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(img.data_ptr()), ctypes.c_void_p(rsd.data_ptr()), ctypes.c_void_p(ru.data_ptr()),
                    ctypes.c_void_p(bias.data_ptr() if bias is not None else 0), ctypes.c_void_p(res.data_ptr() if res is not None else 0),
                    ctypes.c_void_p(y.data_ptr()), C, T, B, K, ctypes.c_float(slope), ctypes.c_float(x_scale),
                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    torch.cuda.synchronize()
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
    """mantissas that maximise the residuals of a bf16 split AND of an 11-bit split (bits 12.. set around the rounding points)"""
    n = int(np.prod(shape))
    mant = (rng.integers(0, 128, n) << 16) | (rng.choice([0x7F, 0x80, 0x7E, 0x81, 0x0F, 0x10, 0x17, 0x08], n) << 8) | rng.choice([0x7F, 0x80, 0xFF, 0x01], n)
    return f32_from_bits(rng.integers(0, 2, n), rng.integers(e_lo, e_hi + 1, n), mant).reshape(shape)


def conv64(x, w, K):
    return F.conv1d(x.double(), w.double(), None, padding=(K - 1) // 2)


def fp32_conv_err(x, w, K, want, scale):
    y = F.conv1d(x, w, None, padding=(K - 1) // 2).double()
    return float(((y - want).abs() / scale).max())


print("== accuracy: max |err| / sum|w x| against an fp64 conv (2^-20 = %.2e; exact-fp32 MFMA path on these: ~5.3e-7) ==" % 2.0 ** -20)
for name, C, K, T in (("max-residual operands", 128, 11, 260), ("max-residual operands", 128, 7, 400), ("max-residual operands", 128, 3, 300),
                      ("randn", 256, 11, 300), ("alternating-sign cancellation", 256, 11, 300), ("wide range 1e-5..1e2 in one tile", 128, 11, 300),
                      ("weight-norm g 1e-2..1e1 per row", 128, 7, 300)):
    rng = np.random.default_rng(C + K + T)
    if name.startswith("max"):
        x = max_residual_values(rng, (1, C, T), 120, 130)
        w = max_residual_values(rng, (C, C, K), 115, 122)
    elif name == "randn":
        x = torch.randn(1, C, T, generator=torch.Generator().manual_seed(1))
        w = torch.randn(C, C, K, generator=torch.Generator().manual_seed(2)) / np.sqrt(C * K)
    elif name.startswith("alternating"):
        sign = np.where((np.arange(C)[:, None] + np.arange(T)[None, :]) % 2 == 0, 1.0, -1.0)
        x = torch.from_numpy((sign * (1.0 + 1e-3 * rng.standard_normal((C, T)))).astype(np.float32))[None]
        w = torch.from_numpy((0.05 * (1.0 + 1e-3 * rng.standard_normal((C, C, K)))).astype(np.float32))
    elif name.startswith("wide"):
        mag = 10.0 ** rng.uniform(-5, 2, (1, C, T))
        x = torch.from_numpy((mag * rng.choice([-1.0, 1.0], (1, C, T))).astype(np.float32))
        w = torch.randn(C, C, K, generator=torch.Generator().manual_seed(2)) / np.sqrt(C * K)
    else:
        x = torch.randn(1, C, T, generator=torch.Generator().manual_seed(1)) * 3
        g = torch.from_numpy(10.0 ** rng.uniform(-2, 1, (C, 1, 1))).float()
        v = torch.randn(C, C, K, generator=torch.Generator().manual_seed(2))
        w = g * v / v.flatten(1).norm(dim=1)[:, None, None]
    want = conv64(x, w, K)
    scale = conv64(x.abs(), w.abs(), K)
    img, rs = pack_h2(w)
    yh = h2(x.to(dev), img, rs, None, None, K).cpu().double()
    yd = torch.empty(1, C, T, device=dev)
    ops.conv1d(ops.PackedConv(w, None, dev), x.to(dev), yd)
    eh = float(((yh - want).abs() / scale).max())
    ed = float(((yd.cpu().double() - want).abs() / scale).max())
    e32 = fp32_conv_err(x, w, K, want, scale)
    rel = float((yh - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
    print("%-36s C=%3d k=%2d: three fp16 products %.3e   six bf16 (library) %.3e   torch fp32 CPU conv %.3e   rel RMS %.2e" % (name, C, K, eh, ed, e32, rel))

print("== fp16 denormal inputs on the matrix pipe: activations of 1e-6 (hi part below fp16's normal range, x_scale = 1) ==")
C, K, T = 128, 3, 300
x = torch.randn(1, C, T, generator=torch.Generator().manual_seed(3)) * 1e-6
w = torch.randn(C, C, K, generator=torch.Generator().manual_seed(4)) / np.sqrt(C * K)
want = conv64(x, w, K)
scale = conv64(x.abs(), w.abs(), K)
img, rs = pack_h2(w)
for xsc in (1.0, 2.0 ** 14):
    yh = h2(x.to(dev), img, rs, None, None, K, x_scale=xsc).cpu().double()
    print("x ~ 1e-6, x_scale = 2^%d: max err / sum|wx| = %.3e (all-zero output: %s)" % (int(np.log2(xsc)), float(((yh - want).abs() / scale).max()), bool((yh == 0).all())))

print("== time at the headline shapes (B = 32; lrelu in, bias + residual as in a ResBlock conv) ==")
for C, K, T in ((128, 11, 49280), (256, 11, 6160), (128, 7, 49280), (256, 7, 6160), (256, 3, 6160), (128, 3, 49280)):
    B = 32
    g = torch.Generator().manual_seed(C + K)
    x = torch.randn(B, C, T, generator=g).to(dev)
    res = torch.randn(B, C, T, generator=g).to(dev)
    w = torch.randn(C, C, K, generator=g) / np.sqrt(C * K)
    bias = torch.randn(C, generator=g)
    img, rs = pack_h2(w)
    pc = ops.PackedConv(w, bias, dev)
    y = torch.empty_like(x)
    bd = bias.to(dev)
    ru = (1.0 / rs.double()).float().to(dev)
    rsd = rs.to(dev)
    y2 = torch.empty_like(x)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run_h2():
        H2.h2_conv(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(img.data_ptr()), ctypes.c_void_p(rsd.data_ptr()), ctypes.c_void_p(ru.data_ptr()),
                   ctypes.c_void_p(bd.data_ptr()), ctypes.c_void_p(res.data_ptr()), ctypes.c_void_p(y2.data_ptr()), C, T, B, K,
                   ctypes.c_float(0.1), ctypes.c_float(1.0), st)

    t_d = time_us(lambda: ops.conv1d(pc, x, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=res))
    t_h = time_us(run_h2)
    t_d2 = time_us(lambda: ops.conv1d(pc, x, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=res))
    t_h2 = time_us(run_h2)
    rel = float((y2 - y).double().pow(2).mean().sqrt() / y.double().pow(2).mean().sqrt())
    fl = 2.0 * C * C * K * T * B
    print("C=%3d k=%2d T=%5d: six bf16 %7.1f / %7.1f us (%5.1f TF-eq)   three fp16 %7.1f / %7.1f us (%5.1f TF-eq)   speed-up %.3f   rel diff %.1e"
          % (C, K, T, t_d, t_d2, fl / min(t_d, t_d2) / 1e6, t_h, t_h2, fl / min(t_h, t_h2) / 1e6, min(t_d, t_d2) / min(t_h, t_h2), rel))

print("== the same launches on all-zero operands (no switching activity: full clock) ==")
for C, K, T in ((128, 11, 49280), (128, 7, 49280)):
    B = 32
    x = torch.zeros(B, C, T, device=dev)
    res = torch.zeros(B, C, T, device=dev)
    w = torch.zeros(C, C, K)
    img, rs = pack_h2(w)
    pc = ops.PackedConv(w, torch.zeros(C), dev)
    y = torch.empty_like(x)
    bd = torch.zeros(C, device=dev)
    rsd = rs.to(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run_h2z():
        H2.h2_conv(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(img.data_ptr()), ctypes.c_void_p(rsd.data_ptr()), ctypes.c_void_p(rsd.data_ptr()),
                   ctypes.c_void_p(bd.data_ptr()), ctypes.c_void_p(res.data_ptr()), ctypes.c_void_p(y.data_ptr()), C, T, B, K,
                   ctypes.c_float(0.1), ctypes.c_float(1.0), st)

    t_h = time_us(run_h2z)
    t_d = time_us(lambda: ops.conv1d(pc, x, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=res))
    print("C=%3d k=%2d T=%5d zeros: six bf16 %7.1f us   three fp16 %7.1f us   speed-up %.3f" % (C, K, T, t_d, t_h, t_d / t_h))
