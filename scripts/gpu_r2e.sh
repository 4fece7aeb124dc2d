#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r2e; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_mas_gpu.py tests/test_resblock_gpu.py tests/test_glow_gpu.py -m gpu -q -rf -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Extension" $OUT/pytest.log | tail -15
for B in 32 256; do
  timeout 300 python bench.py --workload mas --mas-batch $B --steps 20 > $OUT/mas_mw_$B.json 2>>$OUT/err.log
  TTSAMD_MAS_SINGLE_WAVE=1 timeout 300 python bench.py --workload mas --mas-batch $B --steps 20 --no-cpu-baseline > $OUT/mas_sw_$B.json 2>>$OUT/err.log
  python -c "
import json
for n in ('mw','sw'):
    d=json.load(open('$OUT/mas_%s_$B.json'%n)); print('B=$B',n, '%.3f ms'%d['ms_per_step'], '%.3g cells/s'%d['value'], 'hbm frac %.4f'%d['roofline']['frac'], d.get('cpu_baseline',{}).get('value'))"
done
tail -3 $OUT/err.log
