#!/bin/bash
# per-shape kernel medians of the headline step (serial branches, one lane): bash scripts/gpu_r4_shapes.sh <tag>
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r04}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
HEAD=$(cat $R/.build_head 2>/dev/null || echo unknown); STAMP=$(cd $R && python -c "import bench; print(bench.code_stamp())" 2>/dev/null)
HDR="# commit $HEAD kernel-source stamp $STAMP ($(date -u +%Y-%m-%dT%H:%MZ), MI355X via gpurun)"
PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $R/bench.py --serial-branches --lanes 1 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-live-pmc > $OUT/prof.log 2>&1; echo "prof rc=$?"
S=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); T=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
{ echo "$HDR"; python $R/scripts/prof_summary.py stats $S; } > $OUT/kernel_stats.txt
{ echo "$HDR"; python $R/scripts/trace_shapes.py $T 16 70; } > $OUT/per_shape.txt
head -45 $OUT/per_shape.txt | cut -c1-150
rm -rf $OUT/prof
