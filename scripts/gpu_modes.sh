#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/modes; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_resblock_gpu.py tests/test_hifigan_gpu.py tests/test_vits_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
for i in 1 2 3; do for m in 0 1; do
echo -n "run $i group $m: "; TTSAMD_GROUP_BRANCHES=$m timeout 200 python scripts/b1_quick.py 2>&1 | grep request
done; done 2>&1 | tee $OUT/group.txt
