"""Shader-clock phases of ONE mid-grid block of a conv launch (debug build only):
  TTSAMD_BUILD_TAG=dbg TTSAMD_EXTRA_FLAGS=-DTTSAMD_PHASE_CLOCKS python -m tts_amd.build
  TTSAMD_LIB_PATH=tts_amd/libtts_amd_dbg.so python scripts/phase_clocks.py "B,Cout,K,D,T,Cin" ...
-> cycles of prologue (first staged chunk) / K loop (+ wave-group reduction) / epilogue issue / store drain, the block's wall
time on the 100 MHz constant clock and the shader clock that implies, next to the launch's HIP-event time."""
import sys

import torch

sys.path.insert(0, ".")
from tts_amd import ops  # noqa: E402


def run(spec):
    v = list(map(int, spec.split(",")))
    B, C, K, D, T, Cin = v[:6]
    mode = v[6] if len(v) > 6 else 0          # 1 = GATE (C = 2 * hidden rows in, hidden rows out)
    dev = "cuda:0"
    pc = ops.PackedConv(torch.randn(C, Cin, K) / (Cin * K) ** 0.5, torch.randn(C), dev, dilation=D)
    x = torch.randn(B, Cin, T, device=dev)
    y = torch.empty(B, C // 2 if mode == 1 else C, T, device=dev)
    y2 = torch.zeros(1, 1, 16, device=dev, dtype=torch.float32)
    if mode == 1:
        f = lambda: ops.conv1d(pc, x, y, mode=ops.CONV_GATE, y2=y2)  # noqa: E731
    else:
        f = lambda: ops.conv1d(pc, x, y, y2=y2)  # noqa: E731
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    p = y2.view(torch.int64).cpu().tolist()[0][0]
    wall_us = p[5] / 100.0
    print("%-26s launch %6.1f us | block: prologue %6d  loop %7d  epilogue %6d  drain %6d cycles = %6.1f us wall -> %.2f GHz"
          % (spec, e0.elapsed_time(e1) * 100, p[1], p[2] - p[1], p[3] - p[2], p[4] - p[3], wall_us,
             p[4] / max(wall_us, 1e-3) / 1e3), flush=True)


for s in sys.argv[1:]:
    run(s)
