#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r2j; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_hifigan_gpu.py tests/test_synth_gpu.py -m gpu -q -rf -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Extension" $OUT/pytest.log | tail -25
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/conv_post.txt
import sys, torch
sys.path.insert(0, '.')
from tts_amd import ops
dev='cuda:0'
for B,C,T in ((32,32,197120),(16,32,2099712)):
    w=torch.randn(1,C,7)/(C*7)**0.5; pc=ops.PackedConv(w, torch.randn(1), dev)
    x=torch.randn(B,C,T,device=dev); y=torch.empty(B,1,T,device=dev)
    f=lambda: ops.conv1d(pc,x,y,in_act=ops.ACT_LRELU,in_slope=0.01,out_act=ops.ACT_TANH)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/10
    byts=4.0*B*T*(C+1)
    print("conv_post B=%d C=%d T=%d: %.1f us, %.0f GB/s = %.3f of 8 TB/s" % (B,C,T,ms*1e3,byts/ms/1e6,byts/ms/1e6/8000))
PY
