#!/bin/bash
# round 6: two sets of staging registers (two chunks of loads in flight) also at k >= 7: deep1 = the eight-wave mid tiles only, deep2 = every tile
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_deep; mkdir -p $OUT; cd $R
for v in deep2; do TTSAMD_LIB_PATH=$R/tts_amd/libtts_amd_$v.so timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_hifigan_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2 | tee $OUT/pytest_$v.txt; done
for rep in 1 2 3; do for v in default deep1 deep2; do L=tts_amd/libtts_amd_$v.so; [ $v = default ] && L=tts_amd/libtts_amd.so
  echo -n "B=1 $v: "; TTSAMD_LIB_PATH=$R/$L timeout 300 python scripts/b1_quick.py 60 2>&1 | grep "p50"
done; done | tee $OUT/b1.txt
for rep in 1 2; do for v in default deep2; do L=tts_amd/libtts_amd_$v.so; [ $v = default ] && L=tts_amd/libtts_amd.so
  echo -n "B=32 $v: "; TTSAMD_LIB_PATH=$R/$L timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done | tee $OUT/b32.txt
