#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-glowgaps}; shift; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o g -- python $R/bench.py --workload glow_hifigan_v2 --steps 12 --warmup 4 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
T=$(find $OUT/tr -name '*kernel_trace.csv' | head -1)
python $R/scripts/req_gaps.py $T > $OUT/gaps.txt; tail -80 $OUT/gaps.txt
rm -rf $OUT/tr
