#!/bin/bash
# round 6: pair-kernel epilogues with the row unscale and the bias in one fma (new) against multiply + add (prev), same box
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_fold; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_resblock_gpu.py tests/test_hifigan_gpu.py tests/test_vits_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2 | tee $OUT/pytest.txt
for v in prev new; do L=tts_amd/libtts_amd.so; [ $v = prev ] && L=tts_amd/libtts_amd_prev.so
  TTSAMD_LIB_PATH=$R/$L timeout 600 python scripts/r6_pairs_ab.py pairs 2>&1 | grep -v amdgpu.ids > $OUT/kernels_$v.txt
done
python - <<'PY' | tee $OUT/kernels_ab.txt
import re
a=[l.rstrip() for l in open("gpurun_out/r6_fold/kernels_prev.txt")]; b=[l.rstrip() for l in open("gpurun_out/r6_fold/kernels_new.txt")]
bad=0; sa=sb=0
for x,y in zip(a,b):
    mx=re.search(r"([\d.]+) us",x); my=re.search(r"([\d.]+) us",y)
    name=re.sub(r"^\S+\s+","",x); name=name[:name.index(mx.group(0))].strip()
    same = x.split()[-1]==y.split()[-1]; bad += (not same); sa+=float(mx.group(1)); sb+=float(my.group(1))
    print("%-28s prev %8.1f us  new %8.1f us  %+5.1f %%  %s" % (name, float(mx.group(1)), float(my.group(1)), 100*(float(my.group(1))/float(mx.group(1))-1), "bitwise equal" if same else "DIGEST DIFFERS"))
print("sum prev %.1f new %.1f (%+.2f %%); digest mismatches: %d" % (sa, sb, 100*(sb/sa-1), bad))
PY
for rep in 1 2 3; do for v in prev new; do L=tts_amd/libtts_amd.so; [ $v = prev ] && L=tts_amd/libtts_amd_prev.so
  echo -n "B=32 $v: "; TTSAMD_LIB_PATH=$R/$L timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done | tee $OUT/bench_ab.txt
