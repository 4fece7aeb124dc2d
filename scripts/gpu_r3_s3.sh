#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/s3; mkdir -p $OUT; cd $R
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x -rf -o faulthandler_timeout=250 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Extension" $OUT/pytest.log | tail -4
TTSAMD_LIB_PATH=tts_amd/libtts_amd_dbg.so timeout 200 python scripts/res_phase.py 32,32,3,1,197120 32,32,3,1,197120,2 32,32,3,5,197120 32,64,3,1,98560 32,32,11,1,197120 32,64,11,1,98560 32,128,3,1,49280 2>&1 | grep -v amdgpu.ids | tee $OUT/res_phase.txt
# the round-2 code under the same stress: does the stall reproduce there?
(cd _old_r2 && for i in 1 2 3; do timeout 200 python -X faulthandler lanes_stress.py 300 2 1 > $OUT/old_lanes_$i.txt 2>&1; echo "old-code lanes stress $i rc=$? $(tail -1 $OUT/old_lanes_$i.txt | cut -c1-200)"; done)
for i in 1 2 3; do timeout 200 python -X faulthandler scripts/lanes_stress.py 300 2 1 > $OUT/new_lanes_$i.txt 2>&1; echo "new-code lanes stress $i rc=$? $(tail -1 $OUT/new_lanes_$i.txt | cut -c1-200)"; done
timeout 300 python scripts/b1_latency.py 1 2>&1 | grep -v amdgpu.ids | tee $OUT/b1_latency.txt
