#!/bin/bash
# Secondary bench lines (BASELINE configs[2], MAS, XTTS vocoder streaming) + a power/clock trace of the headline bench.
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-aux}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 600 python bench.py --workload hifigan_v1 --steps 2 --warmup 1 > $OUT/bench_hifigan_v1.json 2> $OUT/hifigan_v1.err; cat $OUT/bench_hifigan_v1.json
timeout 300 python bench.py --workload mas --steps 20 --warmup 3 > $OUT/bench_mas.json 2> $OUT/mas.err; cat $OUT/bench_mas.json
timeout 300 python bench.py --workload xtts_stream --steps 5 --warmup 2 > $OUT/bench_xtts_stream.json 2> $OUT/xtts.err; cat $OUT/bench_xtts_stream.json
# power / clock while the headline workload runs (rocm-smi sampled twice a second)
( timeout 300 python bench.py --steps 600 --warmup 2 --no-cpu-baseline > $OUT/bench_long.json 2>/dev/null ) &
BP=$!
sleep 35
for i in $(seq 1 20); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor junction\)" | tr -s ' ' | tr '\n' ';'; echo
  sleep 1
done > $OUT/power_clock.txt
wait $BP
cat $OUT/bench_long.json | python -c "import json,sys; d=json.load(sys.stdin); print('long run ms/step', d['ms_per_step'])"
head -3 $OUT/power_clock.txt
