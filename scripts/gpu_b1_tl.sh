#!/bin/bash
# per-kernel timeline of the VITS B=1 request: bash scripts/gpu_b1_tl.sh <outdir>
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-b1tl}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o b1 -- python $R/scripts/b1_trace_target.py 1 8 graph > $OUT/trace.log 2>&1
T=$(find $OUT/tr -name '*kernel_trace.csv' | head -1)
python $R/scripts/b1_timeline.py $T 8 > $OUT/b1_timeline.txt; head -1 $OUT/b1_timeline.txt
rm -rf $OUT/tr
