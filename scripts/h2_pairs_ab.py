"""Same-box A/B of the fused ResBlock pairs and the unfused convs at the headline shapes (B = 32): six bf16 products ("x3")
vs three fp16 products ("h2").   python scripts/h2_pairs_ab.py"""
import sys

import torch

sys.path.insert(0, ".")
from tts_amd import ops  # noqa: E402

dev = "cuda:0"
B = 32


def time_us(f, n=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print("%-34s %10s %10s %8s   %s" % ("launch", "x3 us", "h2 us", "speedup", "h2 TF-eq (frac of 833)  | rel diff"))
for C, T in ((128, 49280), (64, 98560), (32, 197120)):
    for K in (3, 7, 11):
        for D in (1, 5):
            g = torch.Generator().manual_seed(C + K + D)
            x = torch.randn(B, C, T, generator=g).to(dev)
            y, y2 = torch.empty_like(x), torch.empty_like(x)
            pc1 = ops.PackedConv(torch.randn(C, C, K, generator=g) / (C * K) ** 0.5, torch.randn(C, generator=g), dev, dilation=D)
            pc2 = ops.PackedConv(torch.randn(C, C, K, generator=g) / (C * K) ** 0.5, torch.randn(C, generator=g), dev)
            fused = bool(ops.lib().ttsamd_resblock_pair_supported(C, K, D)) and not (C == 128 and K > 3)
            res = {}
            for prec, out in (("x3", y), ("h2", y2)):
                ops.set_conv_precision(prec)
                if fused:
                    f = lambda: ops.resblock_pair(pc1, pc2, x, out, slope=0.1)  # noqa: E731
                else:
                    f = lambda: ops.conv1d(pc1, x, out, in_act=ops.ACT_LRELU, in_slope=0.1, res=x)  # noqa: E731
                res[prec] = time_us(f)
            rel = float((y2 - y).double().pow(2).mean().sqrt() / y.double().pow(2).mean().sqrt())
            fl = (2 if fused else 1) * 2.0 * C * C * K * T * B
            print("%-34s %10.1f %10.1f %8.3f   %6.1f (%.3f) | %.1e" % (("fused pair" if fused else "conv") + " C=%d k=%d d=%d" % (C, K, D),
                                                                    res["x3"], res["h2"], res["x3"] / res["h2"], fl / res["h2"] / 1e6,
                                                                    fl / res["h2"] / 1e6 / 833.3, rel), flush=True)
            del x, y, y2
for C, T in ((256, 6160),):
    for K in (3, 7, 11):
        g = torch.Generator().manual_seed(C + K)
        x = torch.randn(B, C, T, generator=g).to(dev)
        y, y2 = torch.empty_like(x), torch.empty_like(x)
        pc1 = ops.PackedConv(torch.randn(C, C, K, generator=g) / (C * K) ** 0.5, torch.randn(C, generator=g), dev)
        res = {}
        for prec, out in (("x3", y), ("h2", y2)):
            ops.set_conv_precision(prec)
            res[prec] = time_us(lambda: ops.conv1d(pc1, x, out, in_act=ops.ACT_LRELU, in_slope=0.1, res=x))
        fl = 2.0 * C * C * K * T * B
        print("%-34s %10.1f %10.1f %8.3f   %6.1f (%.3f)" % ("conv C=%d k=%d d=1" % (C, K), res["x3"], res["h2"], res["x3"] / res["h2"],
                                                            fl / res["h2"] / 1e6, fl / res["h2"] / 1e6 / 833.3), flush=True)
# polyphase transposed convs (ups[0..3])
for cin, cout, u, T in ((512, 256, 8, 770), (256, 128, 8, 6160), (128, 64, 2, 49280), (64, 32, 2, 98560)):
    g = torch.Generator().manual_seed(cin)
    wt = torch.randn(cin, cout, 2 * u, generator=g) / (cin * 2) ** 0.5
    w, b = ops.convt_polyphase_weight(wt, torch.randn(cout, generator=g), u)
    pc = ops.PackedConv(w, b, dev, pad_left=1)
    x = torch.randn(B, cin, T, generator=g).to(dev)
    y = torch.empty(B, cout, T * u, device=dev)
    res = {}
    for prec in ("x3", "h2"):
        ops.set_conv_precision(prec)
        res[prec] = time_us(lambda: ops.conv1d(pc, x, y, t_out=T + 1, in_act=ops.ACT_LRELU, in_slope=0.1, mode=ops.CONV_SHUFFLE, shuffle_u=u, shuffle_pad=u // 2))
    print("%-34s %10.1f %10.1f %8.3f" % ("convT %d->%d u=%d" % (cin, cout, u), res["x3"], res["h2"], res["x3"] / res["h2"]), flush=True)
