#!/bin/bash
# round 4: full per-kernel timelines of the two single-request chains (configs[0] Glow-TTS + HiFiGAN-v2 sentence; VITS B=1)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r04tr}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_glow -o g -- python $R/bench.py --workload glow_hifigan_v2 --steps 6 --warmup 3 --no-cpu-baseline > $OUT/glow_trace.log 2>&1
T=$(find $OUT/tr_glow -name '*kernel_trace.csv' | head -1)
python $R/scripts/b1_timeline.py $T 9 > $OUT/glow_timeline.txt; head -1 $OUT/glow_timeline.txt
rm -rf $OUT/tr_glow
PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_b1 -o b1 -- python $R/scripts/b1_trace_target.py 1 8 graph > $OUT/b1_trace.log 2>&1
T=$(find $OUT/tr_b1 -name '*kernel_trace.csv' | head -1)
python $R/scripts/b1_timeline.py $T 8 > $OUT/b1_timeline.txt; head -1 $OUT/b1_timeline.txt
rm -rf $OUT/tr_b1
cd $R
python bench.py --workload glow_hifigan_v2 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400 > $OUT/glow_bench.txt; cat $OUT/glow_bench.txt
python scripts/b1_latency.py > $OUT/b1_latency.txt 2>&1; tail -5 $OUT/b1_latency.txt
