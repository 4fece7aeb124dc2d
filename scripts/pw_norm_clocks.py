"""Phase clocks of the fused [norm ->] 1x1 conv -> norm launch (debug build: TTSAMD_EXTRA_FLAGS=-DTTSAMD_PW_CLOCKS
TTSAMD_BUILD_TAG=clk python -m tts_amd.build; TTSAMD_LIB_PATH=tts_amd/libtts_amd_clk.so python scripts/pw_norm_clocks.py)."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from tts_amd import ops  # noqa: E402

dev = "cuda:0"
C, T, B = 192, 257, 1
g = torch.Generator().manual_seed(0)
x = torch.randn(B, C, T, generator=g).to(dev)
pw_w, pw_b = (torch.randn(C, C, generator=g) / 14).to(dev), torch.randn(C, generator=g).to(dev)
g1, b1, g2, b2 = (torch.randn(C, generator=g).to(dev) for _ in range(4))
dw_w, dw_b = torch.randn(C, 3, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
mask = torch.ones(B, T, device=dev)
y = torch.empty_like(x)
names = ["requests issued", "all loads back", "LN1 + act + LDS write", "barrier", "MFMA", "LN2 + store"]
for label, kw in (("no first norm (encoder form)", dict(pre_res=x, out_mask=mask)),
                  ("DDS layer", dict(first=(g1, b1, 1e-5, ops.ACT_GELU), dw_w=dw_w, dw_bias=dw_b, dw_dilation=3, in_mask=mask, act2=ops.ACT_GELU,
                                     post_res=x, out_mask=mask))):
    for _ in range(5):
        ops.pw_norm(x, y, pw_w, pw_b, g2, b2, 1e-5, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.pw_norm(x, y, pw_w, pw_b, g2, b2, 1e-5, **kw)
    e1.record()
    torch.cuda.synchronize()
    print("%s: %.2f us per launch back to back" % (label, e0.elapsed_time(e1) * 20))
    out = np.zeros((64, 16, 8), np.uint64)
    rc = ops.lib().ttsamd_pw_norm_clocks(ctypes.c_void_p(out.ctypes.data))
    assert rc == 0
    c = out[:17].astype(np.int64)
    d = np.diff(c[:, :, :7], axis=2)
    print("  cycles per phase, median over 17 blocks x 16 waves (wave 0 / wave 15 of block 0 beside it):")
    for i, n in enumerate(names):
        print("    %-24s %7.0f   %7d %7d" % (n, np.median(d[:, :, i]), d[0, 0, i], d[0, 15, i]))
    print("    %-24s %7.0f" % ("total", np.median(c[:, :, 6] - c[:, :, 0])))
    print("    block span (first start -> last end over the block's waves): %d" % (c[0, :, 6].max() - c[0, :, 0].min()))
