#!/bin/bash
# round 6: a single request's fused-pair limits again, on the library with the eight-wave mid tile
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_b1fuse2; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_conv_gpu.py -k "matches_torch" -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2 | tee $OUT/pytest.txt
for rep in 1 2 3; do for v in "" "256:0" "256:3" "256:0,128:3" "256:0,128:0"; do
  echo -n "B=1 TTSAMD_FUSE_LIMITS='$v': "; TTSAMD_FUSE_LIMITS=$v timeout 300 python scripts/b1_quick.py 60 2>&1 | grep "p50"
done; done | tee $OUT/b1_fuse.txt
