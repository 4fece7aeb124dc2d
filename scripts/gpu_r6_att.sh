#!/bin/bash
# round 6: 16-query-block attention kernel (attention_v3.h) — parity on all three kernels, same-box timing, phase clocks
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_att; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_text_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/pytest_text.txt
timeout 600 python scripts/att_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/att_ab.txt
if [ -f tts_amd/libtts_amd_dbg.so ]; then for sh in "1 257 2 96" "32 257 2 96" "1 64 2 96"; do TTSAMD_ATT_V3=1 TTSAMD_LIB_PATH=tts_amd/libtts_amd_dbg.so python scripts/att3_phase.py $sh 2>&1 | grep -v amdgpu.ids; done | tee $OUT/att3_phase.txt; fi
