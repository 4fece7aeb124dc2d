#!/bin/bash
# round 6: second instruction diet (NORMAL conv epilogue: bias through a buffer resource, residual / accumulate passes only when the
# operand exists, no row guard on whole 32-row tiles) on top of the first: d0 = before both, cur = after; digests, tests, headline
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_diet2; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/pytest.txt
for rep in 1 2; do for L in d0 cur; do
  [ "$L" = "cur" ] && lib=tts_amd/libtts_amd.so || lib=tts_amd/libtts_amd_$L.so
  TTSAMD_LIB_PATH=$lib timeout 400 python scripts/r6_pairs_ab.py pairs convs ups 2>&1 | grep -v amdgpu.ids
done; done > $OUT/ab.txt
sed 's#r6_diet/ab.txt#r6_diet2/ab.txt#' scripts/gpu_r6_diet.sh | sed -n '/^python - <<.PY./,/^PY$/p' | sed '1d;$d' > /tmp/sum.py; python /tmp/sum.py | tee $OUT/ab_summary.txt
timeout 900 python scripts/bench_ab.py tts_amd/libtts_amd_d0.so tts_amd/libtts_amd.so 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_ab.txt
