"""The vocoder through its model-level C handle (ttsamd_hifigan_forward, graph replay inside the handle) vs the Python-driven
HifiganGenerator (its own hipGraph cache), same box: HiFiGAN-v2 on one sentence's mel (318 frames -> 83 968 samples) and HiFiGAN-v1
on a 16 x 1024-frame batch; per-call wall time (synchronised) and host time inside the call.   python scripts/native_vocoder_ab.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
from tts_amd import synthetic as W  # noqa: E402
from tts_amd.hifigan import HifiganGenerator, NativeHifigan  # noqa: E402

dev = torch.device("cuda:0")


def make(cfg):
    m = HifiganGenerator(80, 1, cfg["resblock_type"], cfg["resblock_dilation_sizes"], cfg["resblock_kernel_sizes"], cfg["upsample_kernel_sizes"],
                         cfg["upsample_initial_channel"], cfg["upsample_factors"], inference_padding=cfg["inference_padding"])
    sd = W.make_hifigan_state(cfg, 80, seed=1234)
    m.load_state_dict(sd)
    m.to(dev)
    return m, NativeHifigan(m, sd)


def bench(f, n):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    wall, host = [], []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) * 1e6)
        host.append((t1 - t0) * 1e6)
    wall.sort()
    host.sort()
    return wall[n // 2], host[n // 2]


for name, cfg, B, T, n in (("HiFiGAN-v2, one sentence (1 x 318 frames)", dict(W.HIFIGAN_V2), 1, 318, 200), ("HiFiGAN-v1, 16 x 1024 frames", dict(W.HIFIGAN_V1), 16, 1024, 10)):
    m, nat = make(cfg)
    mel = torch.randn(B, 80, T, device=dev)
    out = torch.empty((B, 1, nat.output_samples(T)), device=dev)
    ref = m.inference(mel)
    got = nat.forward(mel, use_graph=True, out=out)
    torch.cuda.synchronize()
    rel = float((got - ref).double().pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt())
    wp, hp = bench(lambda: m.inference(mel), n)
    wn, hn = bench(lambda: nat.forward(mel, use_graph=True, out=out), n)
    we, he = bench(lambda: nat.forward(mel, use_graph=False, out=out), n)
    print("%-44s Python host: %8.1f us per call (host %6.1f)   C handle, graph replay: %8.1f us (host %6.1f)   C handle, eager launches: %8.1f us (host %6.1f)   rel diff %.1e"
          % (name, wp, hp, wn, hn, we, he, rel), flush=True)
    nat.close()
