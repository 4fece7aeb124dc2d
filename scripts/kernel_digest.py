"""Bitwise fingerprint of the conv / fused-ResBlock kernels on fixed seeded inputs: one sha256 per case.
    TTSAMD_LIB_PATH=<variant .so> python scripts/kernel_digest.py > digest_<variant>.txt ; diff the files
A kernel change that is meant to leave the arithmetic alone (staging pipeline, instruction diet, tile arrangement) must
leave every line unchanged."""
import hashlib
import sys

import torch

sys.path.insert(0, ".")
from tts_amd import ops  # noqa: E402


def h(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


def main():
    dev = "cuda:0"
    g = torch.Generator().manual_seed(7)
    for (B, Cin, C, K, D, T, res, accum, div, masked) in [
            (2, 256, 256, 11, 1, 1500, True, True, 3.0, False), (2, 128, 128, 7, 3, 2100, True, False, 0.0, True),
            (3, 256, 256, 3, 5, 700, True, True, 0.0, False), (2, 192, 192, 5, 1, 300, False, False, 0.0, True),
            (1, 512, 256, 7, 1, 900, False, False, 0.0, False), (2, 80, 512, 7, 1, 333, False, False, 0.0, False),
            (4, 64, 64, 11, 5, 1000, True, False, 0.0, False), (2, 32, 32, 3, 1, 4097, True, True, 3.0, True)]:
        w = torch.randn(C, Cin, K, generator=g) / (Cin * K) ** 0.5
        pc = ops.PackedConv(w, torch.randn(C, generator=g), dev, dilation=D)
        x = torch.randn(B, Cin, T, generator=g).to(dev)
        y = torch.empty(B, C, T, device=dev)
        r = torch.randn(B, C, T, generator=g).to(dev) if res else None
        a = torch.randn(B, C, T, generator=g).to(dev) if accum else None
        m = (torch.arange(T)[None, :] < torch.tensor([T - 37 * i for i in range(B)])[:, None]).float().to(dev) if masked else None
        ops.conv1d(pc, x, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=r, accum=a, out_div=div, in_mask=m)
        print("conv   B%d %d->%d k%d d%d T%d res%d acc%d div%g mask%d  %s" % (B, Cin, C, K, D, T, res, accum, div, masked, h(y)))
        ops.conv1d(pc, x, y)                       # no activation (slope-1 identity path)
        print("conv   same, no act                                      %s" % h(y))
    for (B, C, K, D, T, accum, div, masked) in [(2, 32, 3, 1, 3000, True, 3.0, False), (2, 32, 3, 5, 2999, False, 0.0, True),
                                                (2, 64, 3, 3, 2000, True, 0.0, False), (2, 64, 7, 5, 1500, False, 0.0, True),
                                                (1, 64, 11, 1, 1800, True, 3.0, False), (2, 32, 11, 5, 2100, True, 3.0, True),
                                                (2, 128, 3, 1, 1000, False, 0.0, False), (2, 32, 7, 3, 777, False, 0.0, False)]:
        w1 = torch.randn(C, C, K, generator=g) / (C * K) ** 0.5
        w2 = torch.randn(C, C, K, generator=g) / (C * K) ** 0.5
        pc1 = ops.PackedConv(w1, torch.randn(C, generator=g), dev, dilation=D)
        pc2 = ops.PackedConv(w2, torch.randn(C, generator=g), dev, dilation=1)
        x = torch.randn(B, C, T, generator=g).to(dev)
        y = torch.empty(B, C, T, device=dev)
        a = torch.randn(B, C, T, generator=g).to(dev) if accum else None
        m = (torch.arange(T)[None, :] < torch.tensor([T - 91 * i for i in range(B)])[:, None]).float().to(dev) if masked else None
        ops.resblock_pair(pc1, pc2, x, y, slope=0.1, mask=m, accum=a, out_div=div)
        print("pair   B%d c%d k%d d%d T%d acc%d div%g mask%d                %s" % (B, C, K, D, T, accum, div, masked, h(y)))


if __name__ == "__main__":
    main()
