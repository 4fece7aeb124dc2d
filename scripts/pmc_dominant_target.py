"""rocprofv3 --pmc target of bench.py's in-run traffic measurement: the dominant conv instantiation (ResBlock1 k = 11, d = 1:
128 -> 128 at T = 49 280 and 256 -> 256 at T = 6 160, B = 32, leaky-ReLU in, residual in — the headline step's launches), one
warm-up and two counted launches of each.   rocprofv3 --pmc FETCH_SIZE -- python scripts/pmc_dominant_target.py [h2|x3|f32]"""
import sys

import torch

sys.path.insert(0, ".")
from tts_amd import ops  # noqa: E402

if len(sys.argv) > 1:
    ops.set_conv_precision(sys.argv[1])
dev = "cuda:0"
B, K = 32, 11
for C, T in ((128, 49280), (256, 6160)):
    g = torch.Generator().manual_seed(C)
    x = torch.randn(B, C, T, generator=g).to(dev)
    r = torch.randn(B, C, T, generator=g).to(dev)
    y = torch.empty_like(x)
    pc = ops.PackedConv(torch.randn(C, C, K, generator=g) / (C * K) ** 0.5, torch.randn(C, generator=g), dev)
    for _ in range(3):
        ops.conv1d(pc, x, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=r)
    torch.cuda.synchronize()
    del x, y, r
