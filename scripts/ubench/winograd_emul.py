"""Accuracy gate of the products-per-output experiment, emulated on the CPU before any kernel is written:
1-D minimal filtering F(2,3) on tap triples (16 instead of 22 products per output pair at k = 11, 12 vs 14 at k = 7, 4 vs 6 at
k = 3) against the direct form, both accumulated SEQUENTIALLY in fp32 in the kernels' order (channel-major, tap / transform
minor), both against an fp64 conv, on the adversarial operands of tests/test_conv_gpu.py.  fp32 products are taken as exact
(the split-bf16 scheme's dropped cross terms, <= 2^-24.2 per product, are common to both forms).  Transformed weights are
computed in fp64 and rounded once to fp32 (what a load-time packer would do); transformed inputs are computed in fp32 (what
the staging code would do).   python scripts/ubench/winograd_emul.py"""
import numpy as np

f32 = np.float32


def f32_from_bits(sign, exp, mant):
    bits = (sign.astype(np.uint32) << 31) | (exp.astype(np.uint32) << 23) | mant.astype(np.uint32)
    return bits.view(np.float32).copy()


def max_residual_values(rng, shape, e_lo, e_hi):
    n = int(np.prod(shape))
    hi = rng.integers(0, 128, n)
    mid = rng.choice([0x7F, 0x80, 0x7E, 0x81], n)
    lo = rng.choice([0x7F, 0x80, 0xFF, 0x01], n)
    mant = (hi << 16) | (mid << 8) | lo
    return f32_from_bits(rng.integers(0, 2, n), rng.integers(e_lo, e_hi + 1, n), mant).reshape(shape)


def conv64(x, w, K, D):
    """x [C,T], w [Co,C,K] -> [Co,T] (same padding), float64."""
    C, T = x.shape
    pad = (K - 1) * D // 2
    xp = np.zeros((C, T + 2 * pad))
    xp[:, pad:pad + T] = x
    out = np.zeros((w.shape[0], T))
    for j in range(K):
        out += w[:, :, j].astype(np.float64) @ xp[:, j * D: j * D + T]
    return out


def direct_f32(x, w, K, D):
    C, T = x.shape
    pad = (K - 1) * D // 2
    xp = np.zeros((C, T + 2 * pad), f32)
    xp[:, pad:pad + T] = x
    acc = np.zeros((w.shape[0], T), f32)
    for c in range(C):
        for j in range(K):
            acc = (acc + np.outer(w[:, c, j], xp[c, j * D: j * D + T]).astype(f32)).astype(f32)
    return acc


def winograd_f32(x, w, K, D):
    """F(2,3) on tap triples (a last group of 2 or 1 taps runs direct); outputs pair up as (t, t + D)."""
    C, T = x.shape
    Co = w.shape[0]
    pad = (K - 1) * D // 2
    L = T + 2 * pad + 4 * D
    xp = np.zeros((C, L), f32)
    xp[:, pad:pad + T] = x
    # pair index space: positions s = 2*D*(p // D) + p % D  (first output of pair p), second output s + D
    npair = (T + 1) // 2 + D
    p = np.arange(npair)
    s0 = 2 * D * (p // D) + p % D
    s0 = s0[s0 < T]
    m = [np.zeros((Co, len(s0)), f32) for _ in range(4)]
    direct0 = np.zeros((Co, len(s0)), f32)
    direct1 = np.zeros((Co, len(s0)), f32)
    ngrp = K // 3
    rem = K - 3 * ngrp
    wd = w.astype(np.float64)
    for c in range(C):
        for g in range(ngrp):
            g0, g1, g2 = wd[:, c, 3 * g], wd[:, c, 3 * g + 1], wd[:, c, 3 * g + 2]
            U = [g0.astype(f32), ((g0 + g1 + g2) / 2).astype(f32), ((g0 - g1 + g2) / 2).astype(f32), g2.astype(f32)]
            base = s0 + 3 * g * D
            d0, d1, d2, d3 = (xp[c, base + i * D] for i in range(4))
            V = [(d0 - d2).astype(f32), (d1 + d2).astype(f32), (d2 - d1).astype(f32), (d1 - d3).astype(f32)]
            for i in range(4):
                m[i] = (m[i] + np.outer(U[i], V[i]).astype(f32)).astype(f32)
        for r in range(rem):
            j = 3 * ngrp + r
            direct0 = (direct0 + np.outer(w[:, c, j], xp[c, s0 + j * D]).astype(f32)).astype(f32)
            direct1 = (direct1 + np.outer(w[:, c, j], xp[c, s0 + D + j * D]).astype(f32)).astype(f32)
    y0 = (((m[0] + m[1]).astype(f32) + m[2]).astype(f32) + direct0).astype(f32)
    y1 = (((m[1] - m[2]).astype(f32) - m[3]).astype(f32) + direct1).astype(f32)
    out = np.zeros((Co, T + 2 * D), f32)
    out[:, s0] = y0
    out[:, s0 + D] = y1
    return out[:, :T]


def report(name, x, w, K, D):
    want = conv64(x.astype(np.float64), w, K, D)
    scale = conv64(np.abs(x).astype(np.float64), np.abs(w), K, D)
    ed = np.abs(direct_f32(x, w, K, D).astype(np.float64) - want) / scale
    ew = np.abs(winograd_f32(x, w, K, D).astype(np.float64) - want) / scale
    print("%-44s direct: max %.3e rms %.3e | F(2,3): max %.3e rms %.3e | ratio max %.2f rms %.2f   (2^-20 = %.2e)"
          % (name, ed.max(), np.sqrt((ed ** 2).mean()), ew.max(), np.sqrt((ew ** 2).mean()), ew.max() / ed.max(),
             np.sqrt((ew ** 2).mean() / (ed ** 2).mean()), 2.0 ** -20))


if __name__ == "__main__":
    for C, Co, K, D, T in ((64, 64, 7, 1, 400), (256, 64, 11, 1, 260), (32, 32, 3, 5, 300), (128, 64, 11, 3, 300)):
        rng = np.random.default_rng(C + Co + K + D + T)
        x = max_residual_values(rng, (C, T), 120, 130)
        w = max_residual_values(rng, (Co, C, K), 115, 122)
        report("max-residual operands C=%d k=%d d=%d" % (C, K, D), x, w, K, D)
    C, K, T = 256, 11, 300
    rng = np.random.default_rng(2816)
    sign = np.where((np.arange(C)[:, None] + np.arange(T)[None, :]) % 2 == 0, 1.0, -1.0)
    x = (sign * (1.0 + 1e-3 * rng.standard_normal((C, T)))).astype(f32)
    w = (0.05 * (1.0 + 1e-3 * rng.standard_normal((64, C, K)))).astype(f32)
    report("alternating-sign cancellation K=2816", x, w, K, 1)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((128, 300)).astype(f32)
    w = (rng.standard_normal((64, 128, 11)) / np.sqrt(128 * 11)).astype(f32)
    report("randn C=128 k=11", x, w, 11, 1)
