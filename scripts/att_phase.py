"""Phase clocks of the relative-position attention kernel (debug build):
  TTSAMD_BUILD_TAG=dbg TTSAMD_EXTRA_FLAGS=-DTTSAMD_PHASE_CLOCKS python -m tts_amd.build
  TTSAMD_LIB_PATH=tts_amd/libtts_amd_dbg.so python scripts/att_phase.py [B T heads dk]"""
import ctypes, sys
import torch
sys.path.insert(0, '.')
from tts_amd import ops
from tts_amd._lib import lib
B, T, H, dk = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (32, 257, 2, 96)
dev = 'cuda:0'
qkv = torch.randn(B, 3 * H * dk, T, device=dev)
out = torch.empty(B, H * dk, T, device=dev)
mask = torch.ones(B, T, device=dev)
ek, ev = torch.randn(9, dk, device=dev), torch.randn(9, dk, device=dev)
f = lambda: ops.rel_attention(qkv, out, mask, H, ek, ev, 4)
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print("rel_attention B=%d T=%d H=%d dk=%d: %.1f us per launch" % (B, T, H, dk, e0.elapsed_time(e1) * 100))
if hasattr(lib(), "ttsamd_debug_att_clocks"):
    buf = (ctypes.c_longlong * 8)()
    lib().ttsamd_debug_att_clocks(buf)
    s = list(buf)
    names = ["QK^T", "rel-key band", "softmax", "PV", "rel-value band + store"]
    print("  block phases (shader cycles): " + ", ".join("%s %d" % (n, s[i + 1] - s[i]) for i, n in enumerate(names)))
