#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/b1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for mode in graph eager; do
PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$mode -o b1 -- python $R/scripts/b1_trace_target.py 1 8 $mode > $OUT/trace_$mode.log 2>&1
T=$(find $OUT/tr_$mode -name '*kernel_trace.csv' | head -1)
python $R/scripts/b1_timeline.py $T 8 > $OUT/b1_timeline_$mode.txt; head -1 $OUT/b1_timeline_$mode.txt
rm -rf $OUT/tr_$mode
done
