"""Is a kernel on the power envelope?  Each case is looped for a few seconds while `rocm-smi` is sampled twice a second; the
same binary then runs on ALL-ZERO operands (no data toggling in the matrix pipe / LDS / buses: the guide's DVFS experiment).
A kernel whose time drops and whose clock rises on zeros at the same socket power is limited by the power budget, not by
issue slots or memory latency.
    python scripts/power_probe.py [seconds_per_case=4]
-> per case: launch us, TF-equivalent (fraction of the 416.7 split-bf16 ceiling) or GB/s, mean sclk, mean socket power."""
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
from tts_amd import ops  # noqa: E402


class Smi(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop = [], False

    def run(self):
        while not self.stop:
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
                p = re.search(r"Power \(W\):\s*([\d.]+)", out)
                c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
                if p and c:
                    self.rows.append((float(p.group(1)), float(c.group(1))))
            except Exception:
                pass
            time.sleep(0.4)


def loop(f, seconds):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    smi = Smi()
    smi.start()
    n, t0 = 0, time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            f()
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    smi.stop = True
    smi.join(timeout=3)
    rows = smi.rows[1:] or smi.rows or [(0.0, 0.0)]
    return e0.elapsed_time(e1) * 1e3 / n, sum(r[0] for r in rows) / len(rows), sum(r[1] for r in rows) / len(rows)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    dev = "cuda:0"
    B = 32
    print("%-44s %-7s %10s %9s %7s %9s %9s" % ("case", "data", "launch_us", "TF-eq", "frac", "sclk_MHz", "power_W"))
    for name, C, K, D, T, fused in (("conv 256->256 k11 (dominant, 256-ch stage)", 256, 11, 1, 6160, False),
                                    ("conv 128->128 k11 (dominant, 128-ch stage)", 128, 11, 1, 49280, False),
                                    ("conv 128->128 k7", 128, 7, 1, 49280, False),
                                    ("fused pair C=64 k11", 64, 11, 1, 98560, True),
                                    ("fused pair C=32 k3", 32, 3, 1, 197120, True)):
        for data in ("randn", "zeros"):
            mk = (lambda *s: torch.randn(*s)) if data == "randn" else (lambda *s: torch.zeros(*s))
            w1 = mk(C, C, K) / (C * K) ** 0.5
            pc1 = ops.PackedConv(w1, mk(C), dev, dilation=D)
            x = mk(B, C, T).to(dev)
            y = torch.empty_like(x)
            if fused:
                pc2 = ops.PackedConv(mk(C, C, K) / (C * K) ** 0.5, mk(C), dev, dilation=1)
                f = lambda: ops.resblock_pair(pc1, pc2, x, y, slope=0.1)  # noqa: E731
                flops = 2 * 2.0 * C * C * K * T * B
            else:
                r = mk(B, C, T).to(dev)
                f = lambda: ops.conv1d(pc1, x, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=r)  # noqa: E731
                flops = 2.0 * C * C * K * T * B
            us, pw, clk = loop(f, secs)
            print("%-44s %-7s %10.1f %9.1f %7.3f %9.0f %9.0f" % (name, data, us, flops / us / 1e6, flops / us / 1e6 / 416.7, clk, pw), flush=True)
            del x, y
    # an HBM-streaming kernel for comparison: conv_post (32 -> 1, k = 7)
    pc = ops.PackedConv(torch.randn(1, 32, 7) / 15.0, None, dev)
    x = torch.randn(B, 32, 197120, device=dev)
    y = torch.empty(B, 1, 197120, device=dev)
    us, pw, clk = loop(lambda: ops.conv1d(pc, x, y, in_act=ops.ACT_LRELU, in_slope=0.01, out_act=ops.ACT_TANH), secs)
    gb = 4.0 * B * 197120 * 33 / us / 1e3
    print("%-44s %-7s %10.1f %9s %7.3f %9.0f %9.0f   (%.0f GB/s; frac = of 8 TB/s)" % ("conv_post 32->1 k7 (HBM streaming)", "randn", us, "-", gb / 8000, clk, pw, gb), flush=True)


if __name__ == "__main__":
    main()
