"""A/B of the persistent pipelined conv kernel vs one block per tile: python scripts/pipe_ab.py "B,C,K,D,T[,res]" ..."""
import sys
import torch
sys.path.insert(0, '.')
from tts_amd import ops

def run(spec):
    p = spec.split(',')
    B, C, K, D, T = map(int, p[:5])
    res_flag = len(p) > 5 and p[5] == 'res'
    dev = 'cuda:0'
    w = torch.randn(C, C, K) / (C * K) ** 0.5
    pc = ops.PackedConv(w, torch.randn(C), dev, dilation=D)
    x = torch.randn(B, C, T, device=dev)
    y = torch.empty(B, C, T, device=dev)
    r = torch.randn_like(x) if res_flag else None
    f = lambda: ops.conv1d(pc, x, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=r)
    out = []
    for pipe in (False, True, False, True):
        ops.set_conv_pipeline(pipe)
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n * 1e3)
    flops = 2.0 * C * C * K * T * B
    print("%-28s classic %8.1f %8.1f us | pipelined %8.1f %8.1f us  (%5.1f -> %5.1f TF-eq, x%.2f)"
          % (spec, out[0], out[2], out[1], out[3], flops / min(out[0], out[2]) / 1e6, flops / min(out[1], out[3]) / 1e6,
             min(out[0], out[2]) / min(out[1], out[3])), flush=True)

for s in sys.argv[1:]:
    run(s)
