#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r2h; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_vits_gpu.py tests/test_synth_gpu.py -m gpu -q -rf -x -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Extension" $OUT/pytest.log | tail -25
timeout 300 python scripts/b1_latency.py 1 2>&1 | tee $OUT/b1.txt
timeout 300 python scripts/b1_latency.py 2 2>&1 | tee -a $OUT/b1.txt
