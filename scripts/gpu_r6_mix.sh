#!/bin/bash
# round 6: hi / lo split through v_fma_mixlo/hi_f16 (conv_split2x2) against the convert-back form (libtts_amd_prev.so), same box:
# digests must agree line by line
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_mix; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_resblock_gpu.py tests/test_hifigan_gpu.py tests/test_vits_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/pytest.txt
for v in prev new; do L=tts_amd/libtts_amd.so; [ $v = prev ] && L=tts_amd/libtts_amd_prev.so
  TTSAMD_LIB_PATH=$R/$L timeout 600 python scripts/r6_pairs_ab.py pairs convs ups small 2>&1 | grep -v amdgpu.ids > $OUT/kernels_$v.txt
done
python - <<'PY' | tee $OUT/kernels_ab.txt
import re
a=[l.rstrip() for l in open("gpurun_out/r6_mix/kernels_prev.txt")]; b=[l.rstrip() for l in open("gpurun_out/r6_mix/kernels_new.txt")]
bad=0
for x,y in zip(a,b):
    mx=re.search(r"([\d.]+) us",x); my=re.search(r"([\d.]+) us",y)
    name=re.sub(r"^\S+\s+","",x); name=name[:name.index(mx.group(0))].strip()
    same = x.split()[-1]==y.split()[-1]; bad += (not same)
    print("%-40s prev %8.1f us  new %8.1f us  %+5.1f %%  %s" % (name, float(mx.group(1)), float(my.group(1)), 100*(float(my.group(1))/float(mx.group(1))-1), "bitwise equal" if same else "DIGEST DIFFERS"))
print("digest mismatches:", bad)
PY
for rep in 1 2 3; do for v in prev new; do L=tts_amd/libtts_amd.so; [ $v = prev ] && L=tts_amd/libtts_amd_prev.so
  echo -n "B=32 $v: "; TTSAMD_LIB_PATH=$R/$L timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done | tee $OUT/bench_ab.txt
for v in prev new; do L=tts_amd/libtts_amd.so; [ $v = prev ] && L=tts_amd/libtts_amd_prev.so
  echo -n "B=1 $v: "; TTSAMD_LIB_PATH=$R/$L timeout 300 python scripts/b1_quick.py 60 2>&1 | grep "p50"
done | tee $OUT/b1_ab.txt
