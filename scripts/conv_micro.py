"""Micro-benchmark of single conv launches: python scripts/conv_micro.py "B,C,K,D,T[,res]" ...  -> TFLOP/s, GB/s."""
import sys
import torch
sys.path.insert(0, '.')
from tts_amd import ops

def run(spec):
    p = spec.split(',')
    B, C, K, D, T = map(int, p[:5])
    res_flag = len(p) > 5 and p[5] == 'res'
    Cin = int(p[6]) if len(p) > 6 else C          # "B,Cout,K,D,T,res|nores,Cin"
    dev = 'cuda:0'
    w = torch.randn(C, Cin, K) / (Cin * K) ** 0.5
    pc = ops.PackedConv(w, torch.randn(C), dev, dilation=D)
    x = torch.randn(B, Cin, T, device=dev)
    y = torch.empty(B, C, T, device=dev)
    r = torch.randn_like(x) if res_flag else None
    f = lambda: ops.conv1d(pc, x, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=r)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    flops = 2.0 * C * Cin * K * T * B
    byts = 4.0 * T * B * (Cin + C * (1 + res_flag))
    print("%-28s %8.1f us  %6.1f TFLOP/s  %7.1f GB/s" % (spec, ms * 1e3, flops / ms / 1e9, byts / ms / 1e6), flush=True)

for s in sys.argv[1:]:
    run(s)
