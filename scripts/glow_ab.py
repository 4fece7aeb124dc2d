"""Same-box A/B of the configs[0] sentence (Glow-TTS -> seam -> HiFiGAN-v2, B=1): the three model calls one after the other
vs the Synthesizer's SentencePipeline, alternating, throughput loop (no sync between sentences) and per-sentence latency
(synchronised, waveform left on the device).   python scripts/glow_ab.py [rounds] [steps]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from tts_amd import synthetic as W  # noqa: E402
from tts_amd.audio import AudioProcessor, mel_renorm_device  # noqa: E402
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
ap_t, ap_v = AudioProcessor(), AudioProcessor()
T = 64
x = torch.randint(0, 130, (1, T), generator=torch.Generator().manual_seed(0)).to(dev)
aux = {"x_lengths": torch.tensor([T], device=dev), "durations": (4 + (torch.arange(T) % 3)).float().view(1, T).to(dev)}
pipe = SentencePipeline(glow, voc, ap_t, ap_v)


def unfused():
    o = glow.inference(x, aux)
    return voc.inference(mel_renorm_device(o["model_outputs"].transpose(1, 2), ap_t, ap_v))


def fused():
    return pipe(x, aux)[0]


modes = {"three calls": unfused, "pipeline": fused}
for f in modes.values():
    for _ in range(5):
        f()
torch.cuda.synchronize()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
for r in range(rounds):
    for name, f in modes.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            f()
        torch.cuda.synchronize()
        thr = (time.perf_counter() - t0) / steps * 1e3
        lat = []
        for _ in range(steps):
            t0 = time.perf_counter()
            f()
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
        lat.sort()
        print("round %d %-12s throughput loop %.3f ms/sentence; synchronised latency p50 %.3f  p10 %.3f  p90 %.3f ms"
              % (r, name, thr, lat[len(lat) // 2], lat[len(lat) // 10], lat[(9 * len(lat)) // 10]), flush=True)
