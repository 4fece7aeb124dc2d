#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r2i; mkdir -p $OUT; cd $R
for L in "" _cfg2421; do
  echo "== conv_micro lib$L"; TTSAMD_LIB_PATH=$R/tts_amd/libtts_amd$L.so timeout 300 python scripts/conv_micro.py 32,128,11,1,49280,res 32,128,7,1,49280,res 32,128,3,1,49280,res 32,256,11,1,6160,res 32,256,3,1,6160,res 32,128,11,5,49280,nores 2>&1 | grep -v amdgpu.ids | tee $OUT/conv_micro$L.txt
done
for L in "" _cfg2421; do
  TTSAMD_LIB_PATH=$R/tts_amd/libtts_amd$L.so timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 8 > $OUT/bench$L.json 2>/dev/null
  python -c "
import json; d=json.load(open('$OUT/bench$L.json')); print('lib$L', d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_conv_launches']['ms_per_step'])"
done
