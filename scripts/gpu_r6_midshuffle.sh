#!/bin/bash
# round 6: the mid-size (eight-wave) tile for a lone request's ups[0] (TTSAMD_H2_MID_SHUFFLE)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_midshuffle; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_conv_gpu.py -k "polyphase or matches_torch" -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2 | tee $OUT/pytest.txt
for v in 0 1; do TTSAMD_H2_MID_SHUFFLE=$v python - <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
from tts_amd import ops
dev = "cuda:0"
ops.set_conv_precision("h2")
g = torch.Generator().manual_seed(0)
for cin, cout, u, T in ((512, 256, 8, 770), (256, 128, 8, 1500)):
    wt = torch.randn(cin, cout, 2 * u, generator=g) / (cin * 2) ** 0.5
    w, bb = ops.convt_polyphase_weight(wt, torch.randn(cout, generator=g), u)
    pc = ops.PackedConv(w, bb, dev, pad_left=1)
    x = torch.randn(1, cin, T, generator=g).to(dev); y = torch.empty(1, cout, T * u, device=dev)
    f = lambda: ops.conv1d(pc, x, y, t_out=T + 1, in_act=ops.ACT_LRELU, in_slope=0.1, mode=ops.CONV_SHUFFLE, shuffle_u=u, shuffle_pad=u // 2)
    for _ in range(5): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print("MID_SHUFFLE=%s convT %d->%d u=%d T=%d B=1: %.1f us  checksum %.9e" % (os.environ["TTSAMD_H2_MID_SHUFFLE"], cin, cout, u, T, e0.elapsed_time(e1) * 20, float(y.double().sum())))
PY
done 2>&1 | grep -v amdgpu.ids | tee $OUT/kernel.txt
for rep in 1 2 3; do for v in 0 1; do
  echo -n "B=1 TTSAMD_H2_MID_SHUFFLE=$v: "; TTSAMD_H2_MID_SHUFFLE=$v timeout 300 python scripts/b1_quick.py 60 2>&1 | grep "p50"
done; done | tee $OUT/b1.txt
