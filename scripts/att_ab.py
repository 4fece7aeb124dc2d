"""Same-box timing of the three strip attention kernels (TTSAMD_ATT_V3 / TTSAMD_ATT_V2 pick one per process):
python scripts/att_ab.py  -> one line per (kernel, shape), back-to-back launches on one stream."""
import os, subprocess, sys
CODE = r'''
import sys, torch
sys.path.insert(0, '.')
from tts_amd import ops
dev = 'cuda:0'
for B, T, H, dk, w in [(1, 257, 2, 96, 4), (32, 257, 2, 96, 4), (1, 64, 2, 96, 4), (1, 129, 2, 96, 4), (8, 129, 2, 96, 4), (1, 600, 2, 96, 4), (1, 1000, 2, 32, 0)]:
    torch.manual_seed(0)
    qkv = torch.randn(B, 3 * H * dk, T, device=dev)
    out = torch.empty(B, H * dk, T, device=dev)
    mask = torch.ones(B, T, device=dev)
    ek, ev = (torch.randn(2 * w + 1, dk, device=dev), torch.randn(2 * w + 1, dk, device=dev)) if w else (None, None)
    f = lambda: ops.rel_attention(qkv, out, mask, H, ek, ev, w)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(5):
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 50)
    print("%-4s B=%-3d T=%-5d H=%d dk=%-3d window=%d  %7.1f us  checksum %.9e" % (sys.argv[1], B, T, H, dk, w, best, out.double().abs().sum().item()), flush=True)
'''
for name, env in [("v1", {"TTSAMD_ATT_V3": "0", "TTSAMD_ATT_V2": "0"}), ("v2", {"TTSAMD_ATT_V3": "0", "TTSAMD_ATT_V2": "1"}), ("v3", {"TTSAMD_ATT_V3": "1"})]:
    if os.environ.get("TTSAMD_ATT_ONLY", name) != name: continue
    subprocess.run([sys.executable, "-c", CODE, name], env=dict(os.environ, **env))
