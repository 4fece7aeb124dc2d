"""Where does the host spend the time between "the output extent is known" and "the tail graph is launched"?
Wraps ops.durations (returns once the host has the extent) and GraphedSegment.__call__ (staging copy + hipGraphLaunch)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from tts_amd import graphs, ops, synthetic as W  # noqa: E402
from tts_amd.audio import AudioProcessor  # noqa: E402
from tts_amd.glow_tts import GlowTTS  # noqa: E402
from tts_amd.hifigan import HifiganGenerator  # noqa: E402
from tts_amd.synthesizer import SentencePipeline  # noqa: E402

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
ev = []
_dur = ops.durations


def dur(*a, **k):
    ev.append(("dur_in", time.perf_counter()))
    r = _dur(*a, **k)
    ev.append(("dur_out", time.perf_counter()))
    return r


ops.durations = dur
import tts_amd.glow_tts as G  # noqa: E402

G.ops.durations = dur
_call = graphs.GraphedSegment.__call__


def call(self, *inputs):
    ev.append(("seg_in", time.perf_counter()))
    r = _call(self, *inputs)
    ev.append(("seg_out", time.perf_counter()))
    return r


graphs.GraphedSegment.__call__ = call
for _ in range(10):
    pipe(x, aux)
torch.cuda.synchronize()
rows = []
for _ in range(200):
    del ev[:]
    t0 = time.perf_counter()
    pipe(x, aux)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    d = dict()
    segs = [t for n, t in ev if n == "seg_in"], [t for n, t in ev if n == "seg_out"]
    din = [t for n, t in ev if n == "dur_in"][0]
    dout = [t for n, t in ev if n == "dur_out"][0]
    rows.append(((segs[0][0] - t0), (segs[1][0] - segs[0][0]), (din - segs[1][0]), (dout - din), (segs[0][1] - dout), (segs[1][1] - segs[0][1]),
                 (t1 - segs[1][1]), (t2 - t1), t2 - t0))
names = ["entry -> front replay", "front replay call", "front done -> durations call", "durations launch + host wait",
         "extent known -> tail replay call", "tail replay call (staging + hipGraphLaunch)", "tail launched -> return", "return -> device idle", "total"]
for i, n in enumerate(names):
    v = sorted(r[i] for r in rows)
    print("%-48s p50 %7.1f us   p10 %7.1f   p90 %7.1f" % (n, v[100] * 1e6, v[20] * 1e6, v[180] * 1e6))
