#!/bin/bash
# Reproduce the request-lanes stall seen once in round 2: the whole GPU suite in a loop with pytest's
# faulthandler dump armed, so that a stalled run leaves the Python stack of the blocked call under gpurun_out/.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_stall_hunt.sh [runs]'
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/stall; mkdir -p $OUT; cd $R
N=${1:-6}
for i in $(seq 1 $N); do
    timeout 200 python -X faulthandler -m pytest tests -m gpu -q -x -o faulthandler_timeout=120 -p no:cacheprovider > $OUT/run_$i.txt 2>&1
    echo "run $i rc=$? $(tail -1 $OUT/run_$i.txt)"
done
# the two-lane test alone, many times (isolated it passed 2/2)
for i in $(seq 1 10); do
    timeout 60 python -X faulthandler -m pytest tests/test_vits_gpu.py -k "lanes or tail_graph" -q -o faulthandler_timeout=30 -p no:cacheprovider > $OUT/lanes_$i.txt 2>&1
    echo "lanes $i rc=$? $(tail -1 $OUT/lanes_$i.txt)"
done
