#!/bin/bash
# Round artefacts in one GPU call: pytest -m gpu, smoke, bench (x3 + f32), rocprofv3 kernel stats + per-shape table.
# usage: scripts/gpu_round.sh <tag> [notests]
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-round}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
if [ "${2:-}" != "notests" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
fi
timeout 600 python bench.py > $OUT/bench_x3.json 2> $OUT/bench_x3.err; echo "bench rc=$?"; cat $OUT/bench_x3.json
timeout 300 python bench.py --precision f32 --no-cpu-baseline > $OUT/bench_f32.json 2> $OUT/bench_f32.err; cat $OUT/bench_f32.json
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $R/bench.py --serial-branches --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof.log 2>&1
echo "prof rc=$?"
S=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); T=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
python $R/scripts/prof_summary.py stats $S > $OUT/kernel_stats.txt; head -12 $OUT/kernel_stats.txt
python $R/scripts/trace_shapes.py $T 7 70 > $OUT/per_shape.txt; head -40 $OUT/per_shape.txt
rm -rf $OUT/prof
