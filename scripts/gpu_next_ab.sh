#!/bin/bash
# First GPU call of the next round: the experiments prepared (not measured) at the end of round 2.
#   1. attention study (scripts/ubench/att_v2.hip): fp64 check + timing against the library kernel
#   2. every library variant of scripts/build_variants.sh through the conv / resblock / hifigan parity tests
#   3. same-box A/B of the variants on the headline bench line; B = 1 latency for the x3sall variant
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/next_ab; mkdir -p $OUT; cd $R
timeout 120 scripts/ubench/att_v2 2>&1 | tee $OUT/att_v2.txt
for v in pairs noslp pairsns x3sall; do
    L=tts_amd/libtts_amd_$v.so; [ -f $L ] || continue
    TTSAMD_LIB_PATH=$L timeout 400 python -m pytest tests/test_conv_gpu.py tests/test_resblock_gpu.py tests/test_hifigan_gpu.py -m gpu -q -x 2>&1 | tail -2 | sed "s/^/$v: /"
done | tee $OUT/variants_pytest.txt
timeout 900 python scripts/bench_ab.py tts_amd/libtts_amd.so tts_amd/libtts_amd_pairs.so tts_amd/libtts_amd_noslp.so tts_amd/libtts_amd_pairsns.so 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_ab.txt
for L in tts_amd/libtts_amd.so tts_amd/libtts_amd_x3sall.so; do echo "== $L"; TTSAMD_LIB_PATH=$L timeout 200 python scripts/b1_latency.py 1 2>&1 | grep "two graphs"; done | tee $OUT/b1_x3sall.txt
