"""Per-phase shader-clock breakdown of one mid-grid block of the fused ResBlock kernel (library built with
TTSAMD_BUILD_TAG=clocks TTSAMD_EXTRA_FLAGS=-DTTSAMD_PHASE_CLOCKS):  python scripts/resblock_phases.py [C,K,D[,variant]] ..."""
import sys

import torch

sys.path.insert(0, ".")
from tts_amd import ops  # noqa: E402

SHAPES = {32: 197120, 64: 98560, 128: 49280}


def run(spec):
    p = [int(v) for v in spec.split(",")]
    C, K, D = p[:3]
    variant = p[3] if len(p) > 3 else 0
    B, T, dev = 32, SHAPES[C], "cuda:0"
    pc1 = ops.PackedConv(torch.randn(C, C, K) / (C * K) ** 0.5, torch.randn(C), dev, dilation=D)
    pc2 = ops.PackedConv(torch.randn(C, C, K) / (C * K) ** 0.5, torch.randn(C), dev, dilation=1)
    x = torch.randn(B, C, T, device=dev)
    y = torch.empty_like(x)
    dbg = torch.zeros(16, dtype=torch.int64, device=dev)
    f = lambda: ops.resblock_pair(pc1, pc2, x, y, slope=0.1, variant=variant, dbg=dbg)  # noqa: E731
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 4 * 1e3
    s = dbg.tolist()
    d = [s[i + 1] - s[i] for i in range(8)]
    print("c%d k%d d%d v%d %8.1f us | bar+split %6d  bar %5d  conv1 %6d  issue+bar %6d  mid-epi %6d  bar %5d  conv2 %6d  out-epi %6d | tile %6d cyc = %.1f us (%.2f GHz)"
          % (C, K, D, variant, us, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], s[8], s[9] / 100.0, s[8] / max(s[9], 1) * 0.1), flush=True)


for spec in sys.argv[1:] or ["32,3,1", "32,11,1", "32,11,5", "64,3,1", "64,3,1,1", "64,11,1", "64,11,1,1", "128,11,1"]:
    run(spec)
