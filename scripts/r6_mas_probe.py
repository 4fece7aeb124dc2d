"""maximum_path on the device at a few row counts (how much the fifth DP wave of T_x = 257 costs: 256 rows = four waves on
four SIMDs, 257 = five waves, two of them on one SIMD), kernel selected by the environment (TTSAMD_MAS_MW=1: round-3 column
step).  Prints the time per call (forward + backtrack) and a digest of the paths."""
import hashlib
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from tts_amd import helpers  # noqa: E402

dev = torch.device("cuda:0")
for B, TX, TY, ragged in [(32, 192, 770, 0), (32, 256, 770, 0), (32, 257, 770, 0), (32, 320, 770, 0), (32, 257, 770, 1),
                          (256, 257, 770, 1), (32, 64, 770, 0), (32, 512, 1540, 0)]:
    rng = np.random.default_rng(0)
    tx = rng.integers(200 * TX // 257, TX + 1, B) if ragged else np.full(B, TX)
    ty = rng.integers(600, TY + 1, B) if ragged else np.full(B, TY)
    tx[0], ty[0] = TX, TY
    mask = ((np.arange(TX)[None, :, None] < tx[:, None, None]) & (np.arange(TY)[None, None, :] < ty[:, None, None]))
    mask_t = torch.from_numpy(mask.astype(np.float32)).to(dev)
    value = torch.randn(B, TX, TY, device=dev, generator=torch.Generator(dev).manual_seed(1))
    for _ in range(5):
        out = helpers.maximum_path(value, mask_t)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0.record()
        for _ in range(20):
            out = helpers.maximum_path(value, mask_t)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    dig = hashlib.md5(out.cpu().numpy().tobytes()).hexdigest()[:12]
    print("maximum_path [%3d,%3d,%4d] %s  %.1f us   paths %s" % (B, TX, TY, "ragged" if ragged else "full  ", best * 1e3, dig), flush=True)
