"""The vocoder layers north_star prices against the HBM roofline, at the headline shapes (B = 32 x 770 frames), each launched on
random operands and then on all-zero operands (no switching activity: full clock): fused ResBlock pair C = 32 k = 3 (d = 1, 5),
ups[3] (64 -> 32, k = 4, stride 2), ups[2] (128 -> 64, k = 4, stride 2), conv_post.  Plain run: HIP-event times and fractions of
8 TB/s on algorithmic bytes; under rocprofv3 --pmc (scripts/gpu_hbm_layers.sh): the counters of the same launches."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from tts_amd import ops  # noqa: E402

dev = "cuda:0"
B = 32
g = torch.Generator().manual_seed(3)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cases = []


def pair(C, K, D, T):
    w1, w2 = (torch.randn(C, C, K, generator=g) / np.sqrt(C * K) for _ in range(2))
    pc1, pc2 = ops.PackedConv(w1, torch.randn(C, generator=g), dev, dilation=D), ops.PackedConv(w2, torch.randn(C, generator=g), dev)
    xr = torch.randn(B, C, T, generator=g).to(dev)
    y = torch.empty_like(xr)
    for name, x in (("data", xr), ("zeros", torch.zeros_like(xr))):
        cases.append(("fused pair C=%d k=%d d=%d %s" % (C, K, D, name), lambda x=x: ops.resblock_pair(pc1, pc2, x, y, slope=0.1),
                      8.0 * C * T * B, 2 * 2.0 * C * C * K * T * B))


def up(Cin, Cout, T):
    wt = torch.randn(Cin, Cout, 4, generator=g) / np.sqrt(Cin * 4)
    wp, bp = ops.convt_polyphase_weight(wt, torch.randn(Cout, generator=g), 2)
    pc = ops.PackedConv(wp, bp, dev, pad_left=wp.shape[2] - 1)
    xr = torch.randn(B, Cin, T, generator=g).to(dev)
    y = torch.empty(B, Cout, 2 * T, device=dev)
    for name, x in (("data", xr), ("zeros", torch.zeros_like(xr))):
        cases.append(("ups %d->%d k=4 u=2 %s" % (Cin, Cout, name),
                      lambda x=x: ops.conv1d(pc, x, y, t_out=T + wp.shape[2] - 1, in_act=ops.ACT_LRELU, in_slope=0.1, mode=ops.CONV_SHUFFLE,
                                             shuffle_u=2, shuffle_pad=1),
                      4.0 * B * T * (Cin + 2 * Cout), 2.0 * Cin * Cout * 4 * T * B))


def post(C, T):
    pc = ops.PackedConv(torch.randn(1, C, 7, generator=g) / np.sqrt(C * 7), torch.randn(1, generator=g), dev)
    xr = torch.randn(B, C, T, generator=g).to(dev)
    y = torch.empty(B, 1, T, device=dev)
    for name, x in (("data", xr), ("zeros", torch.zeros_like(xr))):
        cases.append(("conv_post C=%d %s" % (C, name), lambda x=x: ops.conv1d(pc, x, y, in_act=ops.ACT_LRELU, in_slope=0.01, out_act=ops.ACT_TANH),
                      4.0 * B * T * (C + 1), 2.0 * C * 7 * T * B))


pair(32, 3, 1, 197120)
pair(32, 3, 5, 197120)
up(64, 32, 98560)
up(128, 64, 49280)
post(32, 197120)
for name, f, nbytes, flops in cases:
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / REPS * 1e3
    print("%-32s %8.1f us  %6.3f of 8 TB/s on algorithmic bytes (%.2f GB)  %6.1f TF-eq = %.3f of 416.7" % (
        name, us, nbytes / us / 1e6 / 8.0, nbytes / 1e9, flops / us / 1e6, flops / us / 1e6 / 416.7), flush=True)
