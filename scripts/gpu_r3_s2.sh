#!/bin/bash
# Round-3 GPU session 2: full suite on the current code, bitwise digests vs the round-2 library, fused pairs / conv shapes
# A/B vs round 2, the bench line, B=1 latency.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/s2; mkdir -p $OUT; cd $R
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x -rf -o faulthandler_timeout=250 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Extension" $OUT/pytest.log | tail -8
for L in r2 ""; do lib=tts_amd/libtts_amd${L:+_$L}.so; TTSAMD_LIB_PATH=$lib timeout 200 python scripts/kernel_digest.py 2>&1 | grep -v amdgpu.ids > $OUT/digest_${L:-new}.txt; done
echo "digest lines differing r2 vs new: $(diff $OUT/digest_r2.txt $OUT/digest_new.txt | grep -c '^<')"
for L in r2 ""; do lib=tts_amd/libtts_amd${L:+_$L}.so; echo "== $lib"; TTSAMD_LIB_PATH=$lib timeout 400 python scripts/resblock_ab.py 2>&1 | grep -v amdgpu.ids; done | tee $OUT/resblock_ab.txt
SH="32,256,11,1,6160,res 32,128,11,1,49280,res 32,256,7,1,6160,res 32,128,7,1,49280,res 32,256,3,1,6160,res 32,128,5,1,49280 32,256,2,1,6160"
for rep in 1 2; do for L in r2 ""; do lib=tts_amd/libtts_amd${L:+_$L}.so; echo "== $lib (pass $rep)"; TTSAMD_LIB_PATH=$lib timeout 200 python scripts/conv_micro.py $SH 2>&1 | grep -v amdgpu.ids; done; done | tee $OUT/conv_ab.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_n1.jsonl 2> $OUT/bench_n1.err; echo "bench rc=$?"; tail -1 $OUT/bench_n1.jsonl | cut -c1-400; awk '{print length($0)}' $OUT/bench_n1.jsonl | tr '\n' ' '; echo
timeout 300 python scripts/b1_latency.py 1 2>&1 | grep -v amdgpu.ids | tee $OUT/b1_latency.txt
