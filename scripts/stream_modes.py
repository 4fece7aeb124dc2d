"""Probe: is the single-sentence time a property of the process or of the HIP stream (hardware queue) the graphs replay on?
One process, the same SentencePipeline replayed on the null stream and on N dedicated streams in turn, 3 passes."""
import sys
import time

import torch

sys.path.insert(0, ".")
from tts_amd import _lib, synthetic as W  # noqa: E402
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
prio = int(sys.argv[2]) if len(sys.argv) > 2 else 0
streams = [None] + [_lib.OwnedStream(dev, priority=prio) for _ in range(n)]


def loop(steps):
    for _ in range(5):
        pipe(x, aux)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe(x, aux)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for p in range(3):
    out = []
    for s in streams:
        if s is None:
            out.append(loop(100))
        else:
            s.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s.stream):
                out.append(loop(100))
    print("pass %d  null stream %.3f | dedicated streams %s ms/sentence" % (p, out[0], " ".join("%.3f" % v for v in out[1:])), flush=True)
