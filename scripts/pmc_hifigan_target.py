"""rocprofv3 --pmc target of bench.py's in-run traffic measurement for BASELINE configs[2]: the HiFiGAN-v1 generator on one slab of
mels (MRF branches on one stream), run twice.   rocprofv3 --pmc FETCH_SIZE -- python scripts/pmc_hifigan_target.py [h2|x3|f32] [items] [frames]"""
import sys

import torch

sys.path.insert(0, ".")
from tts_amd import ops  # noqa: E402
from tts_amd import synthetic as W  # noqa: E402
from tts_amd.hifigan import HifiganGenerator  # noqa: E402

if len(sys.argv) > 1:
    ops.set_conv_precision(sys.argv[1])
items = int(sys.argv[2]) if len(sys.argv) > 2 else 4
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
dev = torch.device("cuda:0")
cfg = dict(W.HIFIGAN_V1)
m = HifiganGenerator(80, 1, cfg["resblock_type"], cfg["resblock_dilation_sizes"], cfg["resblock_kernel_sizes"],
                     cfg["upsample_kernel_sizes"], cfg["upsample_initial_channel"], cfg["upsample_factors"],
                     inference_padding=cfg["inference_padding"])
m.load_state_dict(W.make_hifigan_state(cfg, 80, seed=1234))
m.to(dev)
m.concurrent_branches = False
m.use_graphs = False
mel = torch.randn(items, 80, frames, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
for _ in range(2):
    m.inference(mel)
    torch.cuda.synchronize()
