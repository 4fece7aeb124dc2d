#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r2f; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_mas_gpu.py tests/test_glow_gpu.py -m gpu -q -rf -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Extension" $OUT/pytest.log | tail -8
for B in 32 256; do
  timeout 300 python bench.py --workload mas --mas-batch $B --steps 20 --no-cpu-baseline > $OUT/mas_mw_$B.json 2>>$OUT/err.log
  python -c "
import json
d=json.load(open('$OUT/mas_mw_$B.json')); print('B=$B mw2', '%.3f ms'%d['ms_per_step'], '%.3g cells/s'%d['value'])"
done
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $R/bench.py --workload mas --steps 20 --no-cpu-baseline > $OUT/prof.log 2>&1
S=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); python $R/scripts/prof_summary.py stats $S | head -8; rm -rf $OUT/prof
