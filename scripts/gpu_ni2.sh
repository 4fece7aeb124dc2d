#!/bin/bash
# A/B of the small-grid tiles of one utterance's 256-channel decoder stage (TTSAMD_X3S_WIDE = 0 / 1): parity, single-launch
# times, the B = 1 request and the single sentence
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/ni2; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
for m in 0 1; do
  echo "== TTSAMD_X3S_WIDE=$m"
  TTSAMD_X3S_WIDE=$m timeout 200 python scripts/conv_micro.py 1,256,3,1,6160,res 1,256,7,1,6160,res 1,256,11,1,6160,res 1,128,7,1,2624,res 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $OUT/ab.txt
for r in 1 2; do for m in 0 1; do echo -n "round $r wide $m: "; TTSAMD_X3S_WIDE=$m timeout 200 python scripts/b1_quick.py 2>&1 | grep -v amdgpu.ids | head -1;  TTSAMD_X3S_WIDE=$m timeout 200 python scripts/glow_ab.py 1 50 2>&1 | grep pipeline; done; done 2>&1 | tee -a $OUT/ab.txt
