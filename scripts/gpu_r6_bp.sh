#!/bin/bash
# round 6: activation fragments requested a tap ahead in the eight-wave mid-size tile (NI = 1); nobp = the read-then-use form
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_bp; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_hifigan_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2 | tee $OUT/pytest.txt
for v in nobp default; do L=tts_amd/libtts_amd_$v.so; [ $v = default ] && L=tts_amd/libtts_amd.so
TTSAMD_LIB_PATH=$R/$L V=$v python - <<'PY'
import os, sys, torch, hashlib
sys.path.insert(0, ".")
from tts_amd import ops
dev = "cuda:0"
ops.set_conv_precision("h2")
for C, T, K, D in ((256, 6160, 11, 1), (256, 6160, 11, 5), (256, 6160, 7, 1), (256, 6160, 3, 1), (128, 12000, 11, 3)):
    g = torch.Generator().manual_seed(C + K + D)
    x = torch.randn(1, C, T, generator=g).to(dev); y = torch.empty_like(x)
    pc = ops.PackedConv(torch.randn(C, C, K, generator=g) / (C * K) ** 0.5, torch.randn(C, generator=g), dev, dilation=D)
    f = lambda: ops.conv1d(pc, x, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=x)
    for _ in range(10): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print("%-8s conv C=%d k=%d d=%d T=%d B=1: %.1f us  %s" % (os.environ["V"], C, K, D, T, e0.elapsed_time(e1) * 20, hashlib.md5(y.cpu().numpy().tobytes()).hexdigest()[:10]))
PY
done 2>&1 | grep -v amdgpu.ids | tee $OUT/kernel.txt
for rep in 1 2 3; do for v in nobp default; do L=tts_amd/libtts_amd_$v.so; [ $v = default ] && L=tts_amd/libtts_amd.so
  echo -n "B=1 $v: "; TTSAMD_LIB_PATH=$R/$L timeout 300 python scripts/b1_quick.py 80 2>&1 | grep "p50"
done; done | tee $OUT/b1.txt
