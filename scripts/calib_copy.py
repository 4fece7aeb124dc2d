"""HBM counter calibration: a dword-per-lane streaming copy of a known size through ttsamd_replicate_pad(pad=0)
(same 4-byte-per-lane access width as the conv kernel's activation staging and epilogue stores).  Run under
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE; expected bytes are printed."""
import sys
import torch
sys.path.insert(0, '.')
from tts_amd import ops
rows, t = 4096, 131072                      # 2 GiB in, 2 GiB out: well past the 256 MiB Infinity Cache
x = torch.randn(rows, t, device='cuda:0')
y = torch.empty_like(x)
for _ in range(3):
    ops.replicate_pad(x, y, 0)
torch.cuda.synchronize()
print("bytes read per launch", x.numel() * 4, "written", y.numel() * 4)
