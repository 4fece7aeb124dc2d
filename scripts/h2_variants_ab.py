"""Same-box A/B of tile / fusion choices under the three-product arithmetic (B = 32 headline shapes):
  (a) C = 128, k = 7 / 11: two conv launches vs the fused pair kernel (137 KB -> 91 KB LDS image under h2);
  (b) C = 64: the 4-wave / 128-column tile vs the 8-wave / 256-column tile for every kernel size (round 2 chose per k on six products).
python scripts/h2_variants_ab.py"""
import sys

import torch

sys.path.insert(0, ".")
from tts_amd import ops  # noqa: E402

dev = "cuda:0"
B = 32
ops.set_conv_precision("h2")


def time_us(f, n=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def pair(C, K, D, seed):
    g = torch.Generator().manual_seed(seed)
    pc1 = ops.PackedConv(torch.randn(C, C, K, generator=g) / (C * K) ** 0.5, torch.randn(C, generator=g), dev, dilation=D)
    pc2 = ops.PackedConv(torch.randn(C, C, K, generator=g) / (C * K) ** 0.5, torch.randn(C, generator=g), dev)
    return pc1, pc2


print("(a) C = 128: unfused (two convs) vs fused pair, h2")
T = 49280
for K in (7, 11):
    for D in (1, 5):
        pc1, pc2 = pair(128, K, D, K + D)
        x = torch.randn(B, 128, T, device=dev)
        tmp, y, y2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)

        def unfused():
            ops.conv1d(pc1, x, tmp, in_act=ops.ACT_LRELU, in_slope=0.1)
            ops.conv1d(pc2, tmp, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=x)

        tu = time_us(unfused)
        tf = time_us(lambda: ops.resblock_pair(pc1, pc2, x, y2, slope=0.1))
        rel = float((y2 - y).double().pow(2).mean().sqrt() / y.double().pow(2).mean().sqrt())
        print("  k=%2d d=%d: unfused %7.1f us   fused %7.1f us   fused/unfused %.3f   rel diff %.1e" % (K, D, tu, tf, tf / tu, rel), flush=True)
        del x, tmp, y, y2
print("(b) C = 64: default tile vs the alternative (variant 1 flips 4-wave/128-column <-> 8-wave/256-column), h2")
T = 98560
for K in (3, 7, 11):
    for D in (1, 5):
        pc1, pc2 = pair(64, K, D, K + D)
        x = torch.randn(B, 64, T, device=dev)
        y = torch.empty_like(x)
        t0 = time_us(lambda: ops.resblock_pair(pc1, pc2, x, y, slope=0.1))
        t1 = time_us(lambda: ops.resblock_pair(pc1, pc2, x, y, slope=0.1, variant=1))
        print("  k=%2d d=%d: default (%s) %7.1f us   alternative %7.1f us   alt/default %.3f" % (K, D, "8-wave" if K == 11 else "4-wave", t0, t1, t1 / t0), flush=True)
        del x, y

print("(c) C = 128 fused pairs: 4 waves / 64 columns (default since this measurement: two blocks per CU) vs the 8-wave / 128-column tile (variant 1), h2")
T = 49280
for K in (3, 7):
    for D in (1, 5):
        pc1, pc2 = pair(128, K, D, K + D)
        x = torch.randn(B, 128, T, device=dev)
        y, y1 = torch.empty_like(x), torch.empty_like(x)
        t0 = time_us(lambda: ops.resblock_pair(pc1, pc2, x, y, slope=0.1))
        t1 = time_us(lambda: ops.resblock_pair(pc1, pc2, x, y1, slope=0.1, variant=1))
        rel = float((y1 - y).double().pow(2).mean().sqrt() / y.double().pow(2).mean().sqrt())
        print("  k=%2d d=%d: 4-wave/64-col %7.1f us   8-wave/128-col %7.1f us   8-wave / 4-wave %.3f   rel diff %.1e" % (K, D, t0, t1, t1 / t0, rel), flush=True)
        del x, y, y1
