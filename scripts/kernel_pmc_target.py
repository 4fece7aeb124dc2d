"""Target program for PMC passes over single kernels at benchmark shapes (one launch of each after a warm-up):
the fused ResBlock pair (C=32/64, k in 3,11) and the x3 conv at the dominant shape.  python scripts/kernel_pmc_target.py"""
import sys

import torch

sys.path.insert(0, ".")
from tts_amd import ops  # noqa: E402

dev = "cuda:0"
B = 32
for C, T in ((32, 197120), (64, 98560)):
    x = torch.randn(B, C, T, device=dev)
    y = torch.empty_like(x)
    for K in (3, 11):
        pc1 = ops.PackedConv(torch.randn(C, C, K) / (C * K) ** 0.5, torch.randn(C), dev, dilation=1)
        pc2 = ops.PackedConv(torch.randn(C, C, K) / (C * K) ** 0.5, torch.randn(C), dev, dilation=1)
        for _ in range(2):
            ops.resblock_pair(pc1, pc2, x, y, slope=0.1)
    del x, y
for C, T, K in ((128, 49280, 11), (256, 6160, 11), (128, 49280, 3)):
    x = torch.randn(B, C, T, device=dev)
    y, r = torch.empty_like(x), torch.randn_like(x)
    pc = ops.PackedConv(torch.randn(C, C, K) / (C * K) ** 0.5, torch.randn(C), dev)
    for _ in range(2):
        ops.conv1d(pc, x, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=r)
    del x, y, r
torch.cuda.synchronize()
