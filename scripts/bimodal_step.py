"""One fresh process of the configs[0] sentence loop (bench.py's wl_glow_hifigan_v2 timing method), with the clocks sampled
while it runs: which mode is this process in (1.47 or 1.85 ms per sentence), and do sclk / mclk / power differ between the modes?
    python scripts/bimodal_step.py [steps=300] [tag]
Prints ONE line: tag, ms/sentence of the timed loop, per-step p50 / p90 / max, synchronised latency p50, mean sclk / mclk / power."""
import glob
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
from tts_amd import synthetic as W  # noqa: E402
from tts_amd.audio import AudioProcessor  # noqa: E402
from tts_amd.glow_tts import GlowTTS  # noqa: E402
from tts_amd.hifigan import HifiganGenerator  # noqa: E402
from tts_amd.synthesizer import SentencePipeline  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tag = sys.argv[2] if len(sys.argv) > 2 else "-"


class Smi(threading.Thread):
    """sysfs first (no subprocess: does not steal the host core), rocm-smi as the fallback"""

    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop = [], False
        self.sclk_f = (glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk") or [None])[0]
        self.mclk_f = (glob.glob("/sys/class/drm/card*/device/pp_dpm_mclk") or [None])[0]
        self.pow_f = (glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") or glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input") or [None])[0]

    @staticmethod
    def cur(path):
        for ln in open(path).read().splitlines():
            if "*" in ln:
                m = re.search(r"(\d+)Mhz", ln)
                if m:
                    return float(m.group(1))
        return float("nan")

    def run(self):
        while not self.stop:
            try:
                if self.sclk_f:
                    p = float(open(self.pow_f).read()) / 1e6 if self.pow_f else float("nan")
                    self.rows.append((self.cur(self.sclk_f), self.cur(self.mclk_f) if self.mclk_f else float("nan"), p))
                else:
                    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
                    p = re.search(r"Power \(W\):\s*([\d.]+)", out)
                    c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
                    m = re.search(r"mclk clock level: \d+: \((\d+)Mhz\)", out)
                    self.rows.append((float(c.group(1)) if c else float("nan"), float(m.group(1)) if m else float("nan"), float(p.group(1)) if p else float("nan")))
            except Exception:
                pass
            time.sleep(0.05 if self.sclk_f else 0.4)


dev = torch.device("cuda:0")
hcfg = dict(W.HIFIGAN_V2)
glow = GlowTTS({})
glow.load_state_dict(W.make_glow_state({}, seed=4321))
glow.to(dev)
voc = HifiganGenerator(80, 1, hcfg["resblock_type"], hcfg["resblock_dilation_sizes"], hcfg["resblock_kernel_sizes"],
                       hcfg["upsample_kernel_sizes"], hcfg["upsample_initial_channel"], hcfg["upsample_factors"],
                       inference_padding=hcfg["inference_padding"])
voc.load_state_dict(W.make_hifigan_state(hcfg, 80, seed=1234))
voc.to(dev)
T = 64
x = torch.randint(0, 130, (1, T), generator=torch.Generator().manual_seed(0)).to(dev)
aux = {"x_lengths": torch.tensor([T], device=dev), "durations": (4 + (torch.arange(T) % 3)).float().view(1, T).to(dev)}
pipe = SentencePipeline(glow, voc, AudioProcessor(), AudioProcessor())
for _ in range(2):
    pipe(x, aux, eager=True)
torch.cuda.synchronize()
tw, nw = time.perf_counter(), 0
while nw < 5 or time.perf_counter() - tw < 0.6:
    pipe(x, aux)
    nw += 1
torch.cuda.synchronize()
smi = Smi()
smi.start()
res = []
for rep in range(2):
    ts = [time.perf_counter()]
    for _ in range(steps):
        pipe(x, aux)
        ts.append(time.perf_counter())
    torch.cuda.synchronize()
    end = time.perf_counter()
    d = sorted((b - a) * 1e3 for a, b in zip(ts, ts[1:]))
    res.append(((end - ts[0]) / steps * 1e3, d[len(d) // 2], d[int(len(d) * 0.9)], d[-1]))
lat = []
for _ in range(40):
    t0 = time.perf_counter()
    pipe(x, aux)
    torch.cuda.synchronize()
    lat.append((time.perf_counter() - t0) * 1e3)
lat.sort()
smi.stop = True
smi.join(timeout=3)
rows = smi.rows or [(float("nan"),) * 3]
mean = lambda k: sum(r[k] for r in rows) / len(rows)  # noqa: E731
xa = x.data_ptr()
print("BIMODAL %-22s ms/sentence %.3f %.3f | step p50 %.3f p90 %.3f max %.3f | sync latency p50 %.3f min %.3f | sclk %.0f (min %.0f max %.0f) mclk %.0f power %.0f W (%d samples, %s) | cpu %s | x@%x"
      % (tag, res[0][0], res[1][0], res[1][1], res[1][2], res[1][3], lat[20], lat[0], mean(0), min(r[0] for r in rows), max(r[0] for r in rows), mean(1), mean(2),
         len(rows), "sysfs" if smi.sclk_f else "rocm-smi", sorted(os.sched_getaffinity(0))[:4], xa), flush=True)
