"""One library's timings of the fused ResBlock pairs (three-product kernels) at the headline shapes, with an output digest per
shape so that two libraries can be checked for bitwise agreement:
    TTSAMD_LIB_PATH=tts_amd/libtts_amd_<tag>.so python scripts/r6_pairs_ab.py [pairs] [convs] [ups] [small]
Prints one line per launch: name, microseconds (HIP events, 10 launches), TF-eq, digest."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, ".")
from tts_amd import ops  # noqa: E402

dev = "cuda:0"
B = 32
what = set(sys.argv[1:]) or {"pairs", "convs", "ups", "small"}
variant = int(os.environ.get("PAIR_VARIANT", "0"))
tag = os.path.basename(os.environ.get("TTSAMD_LIB_PATH", "libtts_amd.so")) + ("" if not variant else "/v%d" % variant)


def time_us(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def digest(t):
    return hashlib.md5(t.detach().cpu().numpy().tobytes()).hexdigest()[:10]


ops.set_conv_precision("h2")
if "pairs" in what:
    for C, T in ((128, 49280), (64, 98560), (32, 197120)):
        for K in (3, 7, 11):
            for D in (1, 3, 5):
                if C == 128 and K == 11:
                    continue
                g = torch.Generator().manual_seed(C + K + D)
                x = torch.randn(B, C, T, generator=g).to(dev)
                y = torch.empty_like(x)
                pc1 = ops.PackedConv(torch.randn(C, C, K, generator=g) / (C * K) ** 0.5, torch.randn(C, generator=g), dev, dilation=D)
                pc2 = ops.PackedConv(torch.randn(C, C, K, generator=g) / (C * K) ** 0.5, torch.randn(C, generator=g), dev)
                acc = x.roll(1, 0).contiguous() if D == 5 and K > 3 else None
                f = lambda: ops.resblock_pair(pc1, pc2, x, y, slope=0.1, accum=acc, out_div=3.0 if (acc is not None and K == 11) else 0.0, variant=variant)  # noqa: E731
                us = time_us(f)
                fl = 2 * 2.0 * C * C * K * T * B
                by = (2 + (acc is not None)) * 4.0 * B * C * T
                floor = max(fl / 833.3e6, by / 8e6)
                print("%-16s pair C=%-3d k=%-2d d=%d  %8.1f us  %6.1f TF-eq  roofline %.3f  %s" % (tag, C, K, D, us, fl / us / 1e6, floor / us, digest(y)), flush=True)
                del x, y, acc
if "convs" in what:
    for C, T in ((256, 6160), (128, 49280)):
        for K in (3, 7, 11):
            if C == 128 and K < 11:
                continue
            g = torch.Generator().manual_seed(C + K)
            x = torch.randn(B, C, T, generator=g).to(dev)
            y = torch.empty_like(x)
            pc1 = ops.PackedConv(torch.randn(C, C, K, generator=g) / (C * K) ** 0.5, torch.randn(C, generator=g), dev)
            us = time_us(lambda: ops.conv1d(pc1, x, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=x))
            fl = 2.0 * C * C * K * T * B
            floor = max(fl / 833.3e6, 3 * 4.0 * B * C * T / 8e6)
            print("%-16s conv C=%-3d k=%-2d d=1  %8.1f us  %6.1f TF-eq  roofline %.3f  %s" % (tag, C, K, us, fl / us / 1e6, floor / us, digest(y)), flush=True)
            del x, y
    # the flow WaveNet's gate conv (192 -> 384, k = 5) and res/skip 1x1 at T = 770
    for cin, cout, K, mode in ((192, 384, 5, "gate"), (192, 384, 1, "plain")):
        g = torch.Generator().manual_seed(cin + K)
        x = torch.randn(B, cin, 770, generator=g).to(dev)
        w = torch.randn(cout, cin, K, generator=g) / (cin * K) ** 0.5
        if mode == "gate":
            wg, bg = ops.gate_permute(w, torch.randn(cout, generator=g), cout // 2)
            pc = ops.PackedConv(wg, bg, dev)
            y = torch.empty(B, cout // 2, 770, device=dev)
            f = lambda: ops.conv1d(pc, x, y, mode=ops.CONV_GATE)  # noqa: E731
        else:
            pc = ops.PackedConv(w, torch.randn(cout, generator=g), dev)
            y = torch.empty(B, cout, 770, device=dev)
            f = lambda: ops.conv1d(pc, x, y)  # noqa: E731
        try:
            us = time_us(f)
            print("%-16s conv %d->%d k=%d %s T=770  %8.1f us  %s" % (tag, cin, cout, K, mode, us, digest(y)), flush=True)
        except Exception as e:  # noqa: BLE001
            print("%-16s conv %d->%d k=%d %s: %s" % (tag, cin, cout, K, mode, e))
if "ups" in what:
    for cin, cout, u, T in ((512, 256, 8, 770), (256, 128, 8, 6160), (128, 64, 2, 49280), (64, 32, 2, 98560)):
        g = torch.Generator().manual_seed(cin)
        wt = torch.randn(cin, cout, 2 * u, generator=g) / (cin * 2) ** 0.5
        w, bb = ops.convt_polyphase_weight(wt, torch.randn(cout, generator=g), u)
        pc = ops.PackedConv(w, bb, dev, pad_left=1)
        x = torch.randn(B, cin, T, generator=g).to(dev)
        y = torch.empty(B, cout, T * u, device=dev)
        us = time_us(lambda: ops.conv1d(pc, x, y, t_out=T + 1, in_act=ops.ACT_LRELU, in_slope=0.1, mode=ops.CONV_SHUFFLE, shuffle_u=u, shuffle_pad=u // 2))
        fl = 2.0 * cin * cout * 2 * u * T * B
        floor = max(fl / 833.3e6, 4.0 * B * (cin * T + cout * T * u) / 8e6)
        print("%-16s convT %d->%d u=%d  %8.1f us  roofline %.3f  %s" % (tag, cin, cout, u, us, floor / us, digest(y)), flush=True)
        del x, y
if "small" in what:
    # the B = 32 step's sub-two-round launches: flow WaveNet (T = 770) and text encoder (T = 257) convs, res/skip epilogue included
    for cin, cout, K, T, mode in ((192, 384, 5, 770, "gate"), (192, 384, 1, 770, "res_skip"), (192, 192, 1, 770, "plain"), (192, 768, 3, 257, "plain"),
                                  (768, 192, 3, 257, "plain"), (192, 192, 1, 257, "plain"), (192, 576, 1, 257, "plain")):
        g = torch.Generator().manual_seed(cin + K + T)
        x = torch.randn(B, cin, T, generator=g).to(dev)
        w = torch.randn(cout, cin, K, generator=g) / (cin * K) ** 0.5
        if mode == "gate":
            wg, bg = ops.gate_permute(w, torch.randn(cout, generator=g), cout // 2)
            pc = ops.PackedConv(wg, bg, dev)
            y = torch.empty(B, cout // 2, T, device=dev)
            f = lambda: ops.conv1d(pc, x, y, mode=ops.CONV_GATE)  # noqa: E731
        elif mode == "res_skip":
            pc = ops.PackedConv(w, torch.randn(cout, generator=g), dev)
            y = torch.empty(B, cout // 2, T, device=dev)
            y2 = torch.zeros(B, cout // 2, T, device=dev)
            xr = torch.randn(B, cout // 2, T, generator=g).to(dev)
            f = lambda: ops.conv1d(pc, x, y, mode=ops.CONV_RES_SKIP, res=xr, y2=y2, accum=y2, split_row=cout // 2)  # noqa: E731
        else:
            pc = ops.PackedConv(w, torch.randn(cout, generator=g), dev)
            y = torch.empty(B, cout, T, device=dev)
            f = lambda: ops.conv1d(pc, x, y)  # noqa: E731
        try:
            us = time_us(f, 20)
            fl = 2.0 * cin * cout * K * T * B
            print("%-16s small %d->%d k=%d %s T=%d  %8.1f us  %6.1f TF-eq  %s" % (tag, cin, cout, K, mode, T, us, fl / us / 1e6, digest(y)), flush=True)
        except Exception as e:  # noqa: BLE001
            print("%-16s small %d->%d k=%d %s: %s" % (tag, cin, cout, K, mode, e))
