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


def our_card():
    """sysfs directory of the GPU this process runs on (the box may hold eight: pick ours by PCI address), and the others'."""
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
    cards = [c for c in cards if os.path.exists(os.path.join(c, "pp_dpm_sclk"))]
    try:
        pr = torch.cuda.get_device_properties(0)
        want = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        mine = [c for c in cards if want in os.path.realpath(c)]
    except Exception:
        mine = []
    if not mine and len(cards) == 1:
        mine = cards
    return (mine[0] if mine else None), [c for c in cards if not mine or c != mine[0]]


class Smi(threading.Thread):
    """clocks / power / busy of OUR card and the busy percentage + power of the node's other GPUs, from sysfs (no subprocess)"""

    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop = [], False
        self.card, self.others = our_card()
        self.sclk_f = self.card

    @staticmethod
    def cur(path):
        try:
            for ln in open(path).read().splitlines():
                if "*" in ln:
                    m = re.search(r"(\d+)Mhz", ln)
                    if m:
                        return float(m.group(1))
        except Exception:
            pass
        return float("nan")

    @staticmethod
    def num(pattern, scale=1.0):
        try:
            f = glob.glob(pattern)
            return float(open(f[0]).read()) * scale if f else float("nan")
        except Exception:
            return float("nan")

    def run(self):
        while not self.stop and self.card:
            c = self.card
            busy_o = [self.num(o + "/gpu_busy_percent") for o in self.others]
            pow_o = [self.num(o + "/hwmon/hwmon*/power1_average", 1e-6) for o in self.others]
            self.rows.append((self.cur(c + "/pp_dpm_sclk"), self.cur(c + "/pp_dpm_mclk"), self.num(c + "/hwmon/hwmon*/power1_average", 1e-6),
                              self.cur(c + "/pp_dpm_fclk"), self.num(c + "/gpu_busy_percent"),
                              sum(1 for b in busy_o if b == b and b > 20), sum(p for p in pow_o if p == p)))
            time.sleep(0.05)


dev = torch.device("cuda:0")
hcfg = dict(W.HIFIGAN_V2)
glow = GlowTTS({})
glow.load_state_dict(W.make_glow_state({}, seed=4321))
glow.to(dev)
voc = HifiganGenerator(80, 1, hcfg["resblock_type"], hcfg["resblock_dilation_sizes"], hcfg["resblock_kernel_sizes"],
                       hcfg["upsample_kernel_sizes"], hcfg["upsample_initial_channel"], hcfg["upsample_factors"],
                       inference_padding=hcfg["inference_padding"])
if os.environ.get("GLOW_FUSE_MIX") == "0":       # A/B: affine-coupling conv + separate InvConv / ActNorm kernel instead of the MIX epilogue
    glow.decoder.fuse_mix = False
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
try:
    vbios = open(smi.card + "/vbios_version").read().strip()
except Exception:
    vbios = "?"
rows = smi.rows or [(float("nan"),) * 7]
mean = lambda k: sum(r[k] for r in rows) / len(rows)  # noqa: E731
print("BIMODAL %-22s ms/sentence %.3f %.3f | step p50 %.3f p90 %.3f max %.3f | sync latency p50 %.3f min %.3f | OUR card %s: sclk %.0f (min %.0f max %.0f) "
      "mclk %.0f fclk %.0f busy %.0f%% power %.0f W vbios %s | %d other GPUs on the node: busy(>20%%) %.1f of them, their power %.0f W (%d samples)"
      % (tag, res[0][0], res[1][0], res[1][1], res[1][2], res[1][3], lat[20], lat[0], os.path.basename(os.path.dirname(smi.card)) if smi.card else "?",
         mean(0), min(r[0] for r in rows), max(r[0] for r in rows), mean(1), mean(3), mean(4), mean(2), vbios, len(smi.others), mean(5), mean(6), len(rows)), flush=True)
