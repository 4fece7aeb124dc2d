#!/bin/bash
# per-kernel timeline of the configs[0] sentence: bash scripts/gpu_glow_tl.sh <outdir> [extra bench flags]
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-glowtl}; shift; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_glow -o g -- python $R/bench.py --workload glow_hifigan_v2 --steps 6 --warmup 3 --no-cpu-baseline "$@" > $OUT/glow_trace.log 2>&1
T=$(find $OUT/tr_glow -name '*kernel_trace.csv' | head -1)
python $R/scripts/b1_timeline.py $T 9 > $OUT/glow_timeline.txt; head -1 $OUT/glow_timeline.txt
rm -rf $OUT/tr_glow
