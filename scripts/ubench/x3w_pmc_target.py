"""rocprofv3 --pmc target: the F(2,3) kernel and the direct kernel at the dominant shape, three launches each."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.argv = [sys.argv[0]]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "x3w_bench.py")).read().split('print("== accuracy')[0])
C, K, T, B = 128, 11, 49280, 32
g = torch.Generator().manual_seed(1)
x = torch.randn(B, C, T, generator=g).to(dev)
res = torch.randn(B, C, T, generator=g).to(dev)
w = torch.randn(C, C, K, generator=g) / np.sqrt(C * K)
bias = torch.randn(C, generator=g)
img = pack_transformed(w)
pc = ops.PackedConv(w, bias, dev)
y = torch.empty_like(x)
bd = bias.to(dev)
for _ in range(3):
    x3w(x, img, bd, res, K, 0.1)
    ops.conv1d(pc, x, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=res)
torch.cuda.synchronize()
