"""The 256-channel fused pair (three-product kernel, round 6) against the two conv launches it replaces, same process, same
operands: B = 32 (the headline's 256-channel stage, T = 6160) and B = 1.  python scripts/r6_pair256_ab.py"""
import sys
import torch
sys.path.insert(0, ".")
from tts_amd import ops  # noqa: E402

dev = "cuda:0"
ops.set_conv_precision("h2")


def time_us(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


for B in (32, 1):
    T = 6160
    for K in (3, 7, 11):
        for D in (1, 5):
            g = torch.Generator().manual_seed(K + D)
            C = 256
            x = torch.randn(B, C, T, generator=g).to(dev)
            y, y2, tmp = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
            pc1 = ops.PackedConv(torch.randn(C, C, K, generator=g) / (C * K) ** 0.5, torch.randn(C, generator=g), dev, dilation=D)
            pc2 = ops.PackedConv(torch.randn(C, C, K, generator=g) / (C * K) ** 0.5, torch.randn(C, generator=g), dev)
            acc = x.roll(1, 2).contiguous() if D == 5 else None

            def two():
                ops.conv1d(pc1, x, tmp, in_act=ops.ACT_LRELU, in_slope=0.1)
                ops.conv1d(pc2, tmp, y2, in_act=ops.ACT_LRELU, in_slope=0.1, res=x, accum=acc)
            us_p = time_us(lambda: ops.resblock_pair(pc1, pc2, x, y, slope=0.1, accum=acc))
            us_t = time_us(two)
            rel = float(((y - y2).double().pow(2).mean().sqrt() / y2.double().pow(2).mean().sqrt()).item())
            fl = 2 * 2.0 * C * C * K * T * B
            print("B=%-2d pair C=256 k=%-2d d=%d  fused %8.1f us (%5.1f TF-eq)   two launches %8.1f us   ratio %.3f   rel diff %.1e" % (
                B, K, D, us_p, fl / us_p / 1e6, us_t, us_p / us_t, rel), flush=True)
