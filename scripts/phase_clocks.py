"""Per-phase shader-clock breakdown of one mid-grid block of the split-bf16 conv kernel (debug build with
-DTTSAMD_PHASE_CLOCKS, see scripts/gpu_phase.sh):  python scripts/phase_clocks.py "B,C,K,D,T[,res|nores[,Cin]]" ..."""
import sys
import torch
sys.path.insert(0, '.')
from tts_amd import ops

def run(spec):
    p = spec.split(',')
    B, C, K, D, T = map(int, p[:5])
    res_flag = len(p) > 5 and p[5] == 'res'
    Cin = int(p[6]) if len(p) > 6 else C
    dev = 'cuda:0'
    w = torch.randn(C, Cin, K) / (Cin * K) ** 0.5
    pc = ops.PackedConv(w, torch.randn(C), dev, dilation=D)
    x = torch.randn(B, Cin, T, device=dev)
    y = torch.empty(B, C, T, device=dev)
    r = torch.randn_like(y) if res_flag else None
    dbg = torch.zeros(1, 1, 16, device=dev)       # 6 x int64 stamps
    f = lambda: ops.conv1d(pc, x, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=r, y2=dbg)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    s = dbg.view(torch.int64).flatten()[:6].tolist()
    ghz = s[4] / max(s[5], 1) * 0.1
    flops = 2.0 * C * Cin * K * T * B
    print("%-28s %8.1f us %6.1f TF-eq | block: prologue %6d  main %7d  epilogue-issue %6d  drain %6d cyc (%.1f us total, %.2f GHz)"
          % (spec, ms * 1e3, flops / ms / 1e9, s[1], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] / 100.0, ghz), flush=True)

for s in sys.argv[1:]:
    run(s)
