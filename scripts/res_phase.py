"""Shader-clock phases of ONE mid-grid block (wave 0) of the fused ResBlock kernel (debug build only):
  TTSAMD_BUILD_TAG=dbg TTSAMD_EXTRA_FLAGS=-DTTSAMD_PHASE_CLOCKS python -m tts_amd.build
  TTSAMD_LIB_PATH=tts_amd/libtts_amd_dbg.so python scripts/res_phase.py "B,C,K,D,T[,variant]" ...
-> cycles spent in: x stage-in (loads + split + LDS writes) | barrier | conv1 | mid epilogue | barrier | conv2 | epilogue issue |
store drain; the block's wall time on the 100 MHz constant clock and the shader clock that implies; launch time by HIP events."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from tts_amd import _lib, ops  # noqa: E402


def run(spec):
    p = [int(v) for v in spec.split(",")]
    B, C, K, D, T = p[:5]
    variant = p[5] if len(p) > 5 else 0
    dev = "cuda:0"
    pc1 = ops.PackedConv(torch.randn(C, C, K) / (C * K) ** 0.5, torch.randn(C), dev, dilation=D)
    pc2 = ops.PackedConv(torch.randn(C, C, K) / (C * K) ** 0.5, torch.randn(C), dev, dilation=1)
    x = torch.randn(B, C, T, device=dev)
    y = torch.empty_like(x)
    f = lambda: ops.resblock_pair(pc1, pc2, x, y, slope=0.1, variant=variant)  # noqa: E731
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 16)()
    assert getattr(_lib.lib(), 'ttsamd_debug_res_clocks_k%d' % K)(buf) == 0
    c = [buf[i] - buf[0] for i in range(9)]
    wall_us = (buf[14] - buf[15]) / 100.0
    names = ["stage-in", "barrier", "conv1", "mid epilogue", "barrier", "conv2", "epilogue", "drain"]
    print("%-22s launch %7.1f us | " % (spec, e0.elapsed_time(e1) * 100)
          + "  ".join("%s %d" % (n, c[i + 1] - c[i]) for i, n in enumerate(names))
          + " | total %d cycles = %.1f us wall -> %.2f GHz" % (c[8], wall_us, c[8] / max(wall_us, 1e-3) / 1e3), flush=True)


for s in sys.argv[1:]:
    run(s)
