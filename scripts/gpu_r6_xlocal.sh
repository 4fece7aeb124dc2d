#!/bin/bash
# round 6: x-local XCD order for small weight images + half-width tiles for launches of less than two rounds.
# base = the library of commit d40b30f (tts_amd/libtts_amd_base.so); new = this tree; TTSAMD_H2_NARROW_MAX=0 switches the tile rule off.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_xlocal; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_vits_gpu.py tests/test_text_gpu.py tests/test_glow_gpu.py tests/test_hifigan_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/pytest.txt
{
for v in base new0 new; do
  case $v in base) L=tts_amd/libtts_amd_base.so; N=0;; new0) L=tts_amd/libtts_amd.so; N=0;; new) L=tts_amd/libtts_amd.so; N=1024;; esac
  echo "== $v"; TTSAMD_LIB_PATH=$R/$L TTSAMD_H2_NARROW_MAX=$N timeout 300 python scripts/r6_pairs_ab.py ups small 2>&1 | grep -v amdgpu.ids
done
} | tee $OUT/kernels_ab.txt
for rep in 1 2 3; do for v in base new0 new; do
  case $v in base) L=tts_amd/libtts_amd_base.so; N=0;; new0) L=tts_amd/libtts_amd.so; N=0;; new) L=tts_amd/libtts_amd.so; N=1024;; esac
  echo -n "B=32 $v: "; TTSAMD_LIB_PATH=$R/$L TTSAMD_H2_NARROW_MAX=$N timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done | tee $OUT/bench_ab.txt
for v in base new0 new; do
  case $v in base) L=tts_amd/libtts_amd_base.so; N=0;; new0) L=tts_amd/libtts_amd.so; N=0;; new) L=tts_amd/libtts_amd.so; N=1024;; esac
  echo -n "B=1 $v: "; TTSAMD_LIB_PATH=$R/$L TTSAMD_H2_NARROW_MAX=$N timeout 300 python scripts/b1_quick.py 60 2>&1 | grep "p50"
done | tee $OUT/b1_ab.txt
