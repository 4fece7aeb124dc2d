#!/bin/bash
# same-box A/B of library variants: usage  AB_LIBS="prev w64 ''"  bash scripts/ab_libs.sh  (conv shapes + e2e bench)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/ab; mkdir -p $OUT; cd $R
SH=${AB_CONV:-"32,256,11,1,6160,res 32,128,11,1,49280,res 32,256,7,1,6160,res 32,128,7,1,49280,res 32,256,3,1,6160,res 32,512,3,1,770,res 32,128,5,1,49280 32,256,2,1,6160"}
LIBS=""; for L in $AB_LIBS; do [ "$L" = "cur" ] && LIBS="$LIBS tts_amd/libtts_amd.so" || LIBS="$LIBS tts_amd/libtts_amd_$L.so"; done
for lib in $LIBS; do TTSAMD_LIB_PATH=$lib timeout 200 python scripts/kernel_digest.py 2>&1 | grep -v amdgpu.ids | md5sum | sed "s|-|$lib|"; done
for rep in 1 2; do for lib in $LIBS; do echo "== $lib (pass $rep)"; TTSAMD_LIB_PATH=$lib timeout 200 python scripts/conv_micro.py $SH 2>&1 | grep -v amdgpu.ids; done; done | tee $OUT/conv_ab.txt
timeout 900 python scripts/bench_ab.py $LIBS 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_ab.txt
