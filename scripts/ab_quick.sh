#!/bin/bash
# same-box A/B of the current library against tts_amd/libtts_amd_prev.so: bitwise digests, fused pairs, conv shapes, phase clocks
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/ab; mkdir -p $OUT; cd $R
for L in prev ""; do lib=tts_amd/libtts_amd${L:+_$L}.so; TTSAMD_LIB_PATH=$lib timeout 200 python scripts/kernel_digest.py 2>&1 | grep -v amdgpu.ids > $OUT/digest_${L:-new}.txt; done
echo "digest lines differing prev vs new: $(diff $OUT/digest_prev.txt $OUT/digest_new.txt | grep -c '^<') (of $(wc -l < $OUT/digest_new.txt))"
for rep in 1 2; do for L in prev ""; do lib=tts_amd/libtts_amd${L:+_$L}.so; echo "== $lib (pass $rep)"; TTSAMD_LIB_PATH=$lib timeout 400 python scripts/resblock_ab.py $AB_C 2>&1 | grep -v "amdgpu.ids\|^shape"; done; done | tee $OUT/resblock_ab.txt | awk '/^==/{lib=$2" "$3" "$4} /^c/{print lib, $1,$2,$3,$4, "unfused", $5, "fused", $6}' | sort -k4,7 -s | column -t
if [ -n "$AB_CONV" ]; then for rep in 1 2; do for L in prev ""; do lib=tts_amd/libtts_amd${L:+_$L}.so; echo "== $lib (pass $rep)"; TTSAMD_LIB_PATH=$lib timeout 200 python scripts/conv_micro.py $AB_CONV 2>&1 | grep -v amdgpu.ids; done; done | tee $OUT/conv_ab.txt; fi
[ -f tts_amd/libtts_amd_dbg.so ] && TTSAMD_LIB_PATH=tts_amd/libtts_amd_dbg.so timeout 200 python scripts/res_phase.py 32,32,3,1,197120 32,64,3,1,98560 32,32,11,1,197120 32,64,11,1,98560 32,128,3,1,49280 2>&1 | grep -v amdgpu.ids | tee $OUT/res_phase.txt
if [ -n "$AB_BENCH" ]; then timeout 600 python scripts/bench_ab.py tts_amd/libtts_amd_prev.so tts_amd/libtts_amd.so 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_ab.txt; fi
