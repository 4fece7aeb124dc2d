#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r2d; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_resblock_gpu.py tests/test_conv_gpu.py -m gpu -q -rf -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Extension" $OUT/pytest.log | tail -30
timeout 600 python scripts/resblock_ab.py > $OUT/resblock_ab.txt 2>&1; echo "ab rc=$?"; cat $OUT/resblock_ab.txt
TTSAMD_LIB_PATH=$R/tts_amd/libtts_amd_clocks.so timeout 300 python scripts/resblock_phases.py 2>&1 | tee $OUT/phases.txt
timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 8 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_conv_launches'])"
