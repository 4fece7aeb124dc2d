#!/bin/bash
# Round-3 GPU session 1: (1) full GPU suite + smoke on the current code, (2) bitwise fingerprints of the conv / fused kernels
# against the round-2 library, (3) same-box A/B of the staging-pipeline variants and of the round-2 library on the dominant
# conv shapes and the fused ResBlock pairs, (4) the bench line in its new output format, (5) the request-lanes test in a loop.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/s1; mkdir -p $OUT; cd $R
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x -rf -o faulthandler_timeout=250 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Extension" $OUT/pytest.log | tail -8
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
for L in r2 pipe0 ""; do lib=tts_amd/libtts_amd${L:+_$L}.so; [ -f $lib ] || continue; TTSAMD_LIB_PATH=$lib timeout 200 python scripts/kernel_digest.py 2>&1 | grep -v amdgpu.ids > $OUT/digest_${L:-new}.txt; done
echo "digest diff r2 vs new: $(diff $OUT/digest_r2.txt $OUT/digest_new.txt | grep -c '^<') lines differ; pipe0 vs new: $(diff $OUT/digest_pipe0.txt $OUT/digest_new.txt | grep -c '^<')"; diff $OUT/digest_r2.txt $OUT/digest_new.txt | head -6
SH="32,256,11,1,6160,res 32,128,11,1,49280,res 32,256,7,1,6160,res 32,128,7,1,49280,res 32,256,3,1,6160,res 32,128,5,1,49280 32,256,2,1,6160"
for rep in 1 2; do for L in r2 pipe0 pipe1 ""; do lib=tts_amd/libtts_amd${L:+_$L}.so; [ -f $lib ] || continue; echo "== $lib (pass $rep)"; TTSAMD_LIB_PATH=$lib timeout 200 python scripts/conv_micro.py $SH 2>&1 | grep -v amdgpu.ids; done; done | tee $OUT/conv_ab.txt
for L in r2 ""; do lib=tts_amd/libtts_amd${L:+_$L}.so; echo "== $lib"; TTSAMD_LIB_PATH=$lib timeout 400 python scripts/resblock_ab.py 2>&1 | grep -v amdgpu.ids; done | tee $OUT/resblock_ab.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_n1.jsonl 2> $OUT/bench_n1.err; echo "bench rc=$?"; tail -1 $OUT/bench_n1.jsonl | cut -c1-1500; awk '{print length($0)}' $OUT/bench_n1.jsonl | tr '\n' ' '; echo
timeout 600 python scripts/bench_ab.py tts_amd/libtts_amd.so tts_amd/libtts_amd_pipe0.so tts_amd/libtts_amd_pipe1.so 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_ab.txt
timeout 400 python -X faulthandler scripts/lanes_stress.py 200 2 5 > $OUT/lanes_stress.txt 2>&1; echo "lanes stress rc=$?"; tail -3 $OUT/lanes_stress.txt
for i in 1 2 3; do timeout 100 python -X faulthandler -m pytest tests/test_vits_gpu.py -k "lanes" -q -o faulthandler_timeout=60 -p no:cacheprovider > $OUT/lanes_$i.txt 2>&1; echo -n "lanes$i:rc=$? "; done; echo
