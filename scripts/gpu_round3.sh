#!/bin/bash
# Round-3 evidence in one GPU call, every file stamped with the commit and the kernel-source stamp it was recorded with:
# GPU tests, smoke, the bench lines (headline last), MAS, XTTS streaming, B=1 latency, rocprofv3 kernel stats + per-shape
# medians, PMC passes (HBM traffic FETCH x2 + WRITE, matrix-pipe busy, cycles => clock) for the headline step and configs[2],
# a power / clock trace during the B=32 step, phase clocks of the fused ResBlock kernel.
#   git rev-parse HEAD > .build_head; gpurun --timeout 2400 -- 'bash scripts/gpu_round3.sh r03'
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r03}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
HEAD=$(cat .build_head 2>/dev/null || echo unknown); STAMP=$(python -c "import bench; print(bench.code_stamp())" 2>/dev/null)
HDR="# commit $HEAD kernel-source stamp $STAMP ($(date -u +%Y-%m-%dT%H:%MZ), MI355X via gpurun)"
stamp() { f=$1; { echo "$HDR"; cat $f; } > $f.tmp && mv $f.tmp $f; }
timeout 1200 python -m pytest tests -m gpu -q -rf -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Extension" $OUT/pytest.log | tail -3
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_n1.jsonl 2> $OUT/bench_n1.err; echo "bench rc=$?"; tail -1 $OUT/bench_n1.jsonl | cut -c1-700; awk '{print length($0)}' $OUT/bench_n1.jsonl | tr '\n' ' '; echo
for B in 32 256; do timeout 300 python bench.py --workload mas --mas-batch $B --steps 20 2>/dev/null | tail -1 > $OUT/bench_mas_b$B.json; cut -c1-260 $OUT/bench_mas_b$B.json; done
timeout 300 python bench.py --workload xtts_stream --steps 5 2>/dev/null | tail -1 > $OUT/bench_xtts_stream.json; cut -c1-260 $OUT/bench_xtts_stream.json
timeout 300 python scripts/b1_latency.py 1 2>&1 | grep -v amdgpu.ids > $OUT/b1_latency.txt; cat $OUT/b1_latency.txt; stamp $OUT/b1_latency.txt
[ -f tts_amd/libtts_amd_dbg.so ] && { TTSAMD_LIB_PATH=tts_amd/libtts_amd_dbg.so timeout 200 python scripts/res_phase.py 32,32,3,1,197120 32,32,3,5,197120 32,64,3,1,98560 32,32,11,1,197120 32,64,11,1,98560 32,128,3,1,49280 2>&1 | grep -v amdgpu.ids > $OUT/resblock_phase_clocks.txt; stamp $OUT/resblock_phase_clocks.txt; }
timeout 600 python scripts/resblock_ab.py 2>&1 | grep -v amdgpu.ids > $OUT/resblock_fused_vs_unfused.txt; stamp $OUT/resblock_fused_vs_unfused.txt; tail -4 $OUT/resblock_fused_vs_unfused.txt
timeout 600 python scripts/power_probe.py 4 2>&1 | grep -v amdgpu.ids > $OUT/power_probe.txt; stamp $OUT/power_probe.txt; cat $OUT/power_probe.txt
# power / clock while the headline workload runs (rocm-smi sampled once a second)
( timeout 200 python bench.py --steps 400 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_long.jsonl 2>/dev/null ) &
BP=$!
sleep 30
for i in $(seq 1 15); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor junction\)" | tr -s ' ' | tr '\n' ';'; echo
  sleep 1
done > $OUT/power_clock.txt
wait $BP
echo "# long run: $(tail -1 $OUT/bench_long.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f ms/step over %d steps' % (d['ms_per_step'], d['steps']))")" >> $OUT/power_clock.txt; stamp $OUT/power_clock.txt; head -4 $OUT/power_clock.txt | cut -c1-250
cd /tmp && export TMPDIR=/tmp
BENCH="$R/bench.py --serial-branches --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $R/bench.py --serial-branches --lanes 1 --steps 8 --warmup 2 --no-cpu-baseline --no-extras > $OUT/prof.log 2>&1; echo "prof rc=$?"
S=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); T=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
python $R/scripts/prof_summary.py stats $S > $OUT/kernel_stats.txt; stamp $OUT/kernel_stats.txt; head -12 $OUT/kernel_stats.txt | cut -c1-160
python $R/scripts/trace_shapes.py $T 16 70 > $OUT/per_shape.txt; stamp $OUT/per_shape.txt; head -6 $OUT/per_shape.txt
rm -rf $OUT/prof
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 600 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -o p -- python $BENCH > $OUT/pmc$i.log 2>&1; echo "pmc$i rc=$?"
  cp $(find $OUT/pmc$i -name '*counter_collection.csv' | head -1) $OUT/pmc$i.csv 2>/dev/null; rm -rf $OUT/pmc$i
done
(cd $R && python scripts/pmc_round.py $OUT/pmc_dominant_x3.json "conv1d_x3_kernel<11,1,1,4,4,1,0," $OUT/pmc1.csv $OUT/pmc2.csv $OUT/pmc3.csv > $OUT/pmc_table.txt); stamp $OUT/pmc_table.txt; head -24 $OUT/pmc_table.txt
V1="$R/bench.py --workload hifigan_v1 --items 16 --steps 1 --warmup 1 --no-cpu-baseline"
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  PYTHONPATH=$R timeout 600 rocprofv3 --pmc $P --output-format csv -d $OUT/v1$P -o p -- python $V1 > $OUT/v1$P.log 2>&1; echo "v1 $P rc=$?"
  cp $(find $OUT/v1$P -name '*counter_collection.csv' | head -1) $OUT/v1_$P.csv 2>/dev/null; rm -rf $OUT/v1$P
done
(cd $R && python scripts/pmc_round.py $OUT/pmc_hifigan_v1_x3_resblock.json "resblock_pair_x3_kernel" $OUT/v1_FETCH_SIZE.csv $OUT/v1_WRITE_SIZE.csv > $OUT/pmc_table_hifigan_v1.txt); stamp $OUT/pmc_table_hifigan_v1.txt; head -12 $OUT/pmc_table_hifigan_v1.txt
rm -f $OUT/*.csv $OUT/*.log.tmp
