#!/bin/bash
# Round-2 first GPU call: tests, smoke, fused-vs-unfused A/B, LDS-image A/B of the x3 conv, bench (with extras).
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r2a; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 600 python scripts/resblock_ab.py > $OUT/resblock_ab.txt 2>&1; echo "ab rc=$?"; cat $OUT/resblock_ab.txt
for L in "" _interleaved; do
  echo "== conv_micro lib$L"; TTSAMD_LIB_PATH=$R/tts_amd/libtts_amd$L.so timeout 300 python scripts/conv_micro.py 32,128,11,1,49280,res 32,256,11,1,6160,res 32,128,3,1,49280,res 32,64,11,1,98560,res 32,32,11,1,197120,res 32,32,3,1,197120,res 2>&1 | tee $OUT/conv_micro$L.txt
done
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
