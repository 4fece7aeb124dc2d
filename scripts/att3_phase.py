"""Phase clocks of the 16-query-block attention kernel (debug build):
  TTSAMD_BUILD_TAG=dbg TTSAMD_EXTRA_FLAGS=-DTTSAMD_PHASE_CLOCKS python -m tts_amd.build
  TTSAMD_ATT_V3=1 TTSAMD_LIB_PATH=tts_amd/libtts_amd_dbg.so python scripts/att3_phase.py [B T heads dk]"""
import ctypes, sys
import torch
sys.path.insert(0, '.')
from tts_amd import ops
from tts_amd._lib import lib
B, T, H, dk = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (1, 257, 2, 96)
dev = 'cuda:0'
qkv = torch.randn(B, 3 * H * dk, T, device=dev)
out = torch.empty(B, H * dk, T, device=dev)
mask = torch.ones(B, T, device=dev)
ek, ev = torch.randn(9, dk, device=dev), torch.randn(9, dk, device=dev)
junk = torch.empty(64 << 20, device=dev)
f = lambda: ops.rel_attention(qkv, out, mask, H, ek, ev, 4)
for _ in range(3): f()
torch.cuda.synchronize()
names = ["mask / Ev -> LDS", "Q K^T (+ R)", "barrier", "softmax + barrier", "P V", "barrier", "reduce + band + store"]
for cold in (False, True):
    if cold:
        junk.zero_(); qkv.add_(0.0)        # operands last written by another kernel, caches flushed of them
    f(); torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 8)()
    lib().ttsamd_debug_att_clocks(buf)
    s = list(buf)
    print("rel_attention_v3 B=%d T=%d H=%d dk=%d (%s): block (1,0,0) wave 0, shader clocks: " % (B, T, H, dk, "operands rewritten" if cold else "warm")
          + ", ".join("%s %d" % (n, s[i + 1] - s[i]) for i, n in enumerate(names)) + ", total %d" % (s[7] - s[0]))
