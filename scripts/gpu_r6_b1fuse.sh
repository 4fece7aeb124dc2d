#!/bin/bash
# round 6: the new XCD-order cases, then a single request (VITS B = 1) under different fused-pair limits (TTSAMD_FUSE_LIMITS)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_b1fuse; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_conv_gpu.py -k "xcd or polyphase or multi_tile" -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/pytest.txt
for rep in 1 2; do for v in "" "128:11" "128:11,256:11" "128:3" "256:0" "64:7"; do
  echo -n "B=1 TTSAMD_FUSE_LIMITS='$v': "; TTSAMD_FUSE_LIMITS=$v timeout 300 python scripts/b1_quick.py 60 2>&1 | grep "p50"
done; done | tee $OUT/b1_fuse.txt
