import sys, time, torch
sys.path.insert(0, '.')
from oracle import tts_oracle as O, weights as W
from tts_amd.hifigan import HifiganGenerator
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 770
cfg = dict(W.HIFIGAN_V1, inference_padding=0)
sd = O.make_hifigan_state(cfg, 192, seed=7, pre_wn=False, post_wn=False, post_bias=False)
m = HifiganGenerator(192, 1, "1", cfg["resblock_dilation_sizes"], cfg["resblock_kernel_sizes"], cfg["upsample_kernel_sizes"], 512, cfg["upsample_factors"], inference_padding=0, conv_pre_weight_norm=False, conv_post_weight_norm=False, conv_post_bias=False)
m.load_state_dict(sd); m.to('cuda:0')
x = torch.randn(B, 192, T, device='cuda:0')
for _ in range(2): y = m(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 3
e0.record()
for _ in range(n): y = m(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
flop = 2401984.0 * y.numel()
print(f"B={B} T={T} samples={y.numel()} ms={ms:.2f} TFLOP/s={flop/ms/1e9:.1f} samples/s={y.numel()/ms*1e3:.3e} RTFx={y.numel()/22050/(ms/1e3):.0f}")
