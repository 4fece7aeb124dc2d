"""A/B of the fused ResBlock1-iteration kernel against the two conv launches it replaces, at the benchmark's stage shapes
(B=32 VITS decoder: C=32 T=197120, C=64 T=98560, C=128 T=49280):  python scripts/resblock_ab.py [C ...]
Prints per (C, k, d): unfused pair us, fused us (per variant), speed-up, fused TF-equivalent and fraction of the 416.7
TF-eq split-bf16 ceiling, fused algorithmic GB/s (x read once + y written once) and fraction of 8 TB/s."""
import sys

import torch

sys.path.insert(0, ".")
from tts_amd import ops  # noqa: E402

SHAPES = {32: 197120, 64: 98560, 128: 49280}
import os
B = int(os.environ.get("AB_BATCH", "32"))


def timeit(f, n=4):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = "cuda:0"
    cs = [int(a) for a in sys.argv[1:]] or [32, 64, 128]
    print("%-16s %10s %10s %7s %8s %6s %9s %6s" % ("shape", "unfused_us", "fused_us", "x", "TF-eq", "frac", "GB/s", "hbm"))
    for C in cs:
        T = SHAPES[C]
        x = torch.randn(B, C, T, device=dev)
        tmp, y = torch.empty_like(x), torch.empty_like(x)
        for K in (3, 7, 11):
            for D in (1, 5):
                w1 = torch.randn(C, C, K) / (C * K) ** 0.5
                w2 = torch.randn(C, C, K) / (C * K) ** 0.5
                pc1 = ops.PackedConv(w1, torch.randn(C), dev, dilation=D)
                pc2 = ops.PackedConv(w2, torch.randn(C), dev, dilation=1)

                def unfused():
                    ops.conv1d(pc1, x, tmp, in_act=ops.ACT_LRELU, in_slope=0.1)
                    ops.conv1d(pc2, tmp, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=x)

                tu = timeit(unfused)
                flops = 2 * 2.0 * C * C * K * T * B
                byts = 4.0 * C * T * B * 2
                for variant in ((0, 1, 2) if (C == 64 and K == 3) else (0, 1) if C == 64 else (0, 2) if (C == 32 and K == 3) else (0,)):
                    tf = timeit(lambda: ops.resblock_pair(pc1, pc2, x, y, slope=0.1, variant=variant))
                    print("c%d k%d d%d v%d %s %10.1f %10.1f %7.2f %8.1f %6.3f %9.0f %6.3f"
                          % (C, K, D, variant, " " * (4 - len(str(C)) - len(str(K))), tu, tf, tu / tf, flops / tf / 1e6,
                             flops / tf / 1e6 / 416.7, byts / tf / 1e3, byts / tf / 1e3 / 8000.0), flush=True)


if __name__ == "__main__":
    main()
