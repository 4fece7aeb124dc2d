"""Per-step host timestamps of the configs[0] throughput loop (no sync between sentences), after the same preamble as bench.py
(two eager runs for the launch count): where do slow loops lose their time — every step, or a few hiccups?"""
import sys
import time

import torch

sys.path.insert(0, ".")
from tts_amd import synthetic as W  # noqa: E402
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
if len(sys.argv) > 1 and sys.argv[1] == "eager":
    for _ in range(2):
        pipe(x, aux, eager=True)
    torch.cuda.synchronize()
for _ in range(5):
    pipe(x, aux)
torch.cuda.synchronize()
for rep in range(3):
    ts = [time.perf_counter()]
    for _ in range(50):
        pipe(x, aux)
        ts.append(time.perf_counter())
    torch.cuda.synchronize()
    end = time.perf_counter()
    raw = [(b - a) * 1e3 for a, b in zip(ts, ts[1:])]
    print("        slowest step: index %d; graph stats front %s tail %s; allocator segments %d" % (
        raw.index(max(raw)), glow._front.stats, pipe._graph.stats, torch.cuda.memory_stats()["segment.all.allocated"]), flush=True)
    d = sorted(raw)
    print("loop %d: %.3f ms/sentence | per-step host deltas ms: min %.3f p50 %.3f p90 %.3f max %.3f | last sync %.3f"
          % (rep, (end - ts[0]) / 50 * 1e3, d[0], d[25], d[45], d[-1], (end - ts[-1]) * 1e3), flush=True)
    lat = []
    for _ in range(20):
        t0 = time.perf_counter()
        pipe(x, aux)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
    lat.sort()
    print("        synchronised latency p50 %.3f  min %.3f max %.3f" % (lat[10], lat[0], lat[-1]), flush=True)
