#!/bin/bash
# Round artefacts in one GPU call: full GPU tests, smoke, the bench line (with the configs[0] / configs[2] extras), rocprofv3
# kernel stats + per-shape table, PMC passes (HBM traffic, MFMA-busy / clock) of the headline step and of the configs[2]
# vocoder, MAS at B=32 / B=256, small-request latency.   usage: scripts/gpu_round2.sh <tag>
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r02}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -m gpu -q -rf -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Extension" $OUT/pytest.log | tail -6
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1200 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench_n1.json
for B in 32 256; do timeout 300 python bench.py --workload mas --mas-batch $B --steps 20 > $OUT/bench_mas_b$B.json 2>/dev/null; cut -c1-300 $OUT/bench_mas_b$B.json; done
timeout 300 python bench.py --workload xtts_stream --steps 5 > $OUT/bench_xtts_stream.json 2>/dev/null; cut -c1-300 $OUT/bench_xtts_stream.json
timeout 300 python scripts/b1_latency.py 1 2>&1 | grep -v amdgpu.ids > $OUT/b1_latency.txt; cat $OUT/b1_latency.txt
timeout 600 python scripts/resblock_ab.py 2>&1 | grep -v amdgpu.ids > $OUT/resblock_ab.txt; tail -5 $OUT/resblock_ab.txt
cd /tmp && export TMPDIR=/tmp
BENCH="$R/bench.py --serial-branches --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $R/bench.py --serial-branches --lanes 1 --steps 8 --warmup 2 --no-cpu-baseline --no-extras > $OUT/prof.log 2>&1; echo "prof rc=$?"
S=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); T=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
python $R/scripts/prof_summary.py stats $S > $OUT/kernel_stats.txt; head -14 $OUT/kernel_stats.txt
python $R/scripts/trace_shapes.py $T 16 70 > $OUT/per_shape.txt; head -5 $OUT/per_shape.txt
rm -rf $OUT/prof
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 600 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -o p -- python $BENCH > $OUT/pmc$i.log 2>&1; echo "pmc$i rc=$?"
  cp $(find $OUT/pmc$i -name '*counter_collection.csv' | head -1) $OUT/pmc$i.csv 2>/dev/null; rm -rf $OUT/pmc$i
done
python $R/scripts/pmc_round.py $OUT/pmc_dominant_x3.json "conv1d_x3_kernel<11,1,1,4,4,1,0>" $OUT/pmc1.csv $OUT/pmc2.csv $OUT/pmc3.csv > $OUT/pmc_table.txt; cat $OUT/pmc_table.txt
# configs[2]: 16-item launches (the size of the hbm_subset pass; every launch of this run has that size)
V1="$R/bench.py --workload hifigan_v1 --items 16 --steps 1 --warmup 1 --no-cpu-baseline"
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  PYTHONPATH=$R timeout 600 rocprofv3 --pmc $P --output-format csv -d $OUT/v1$P -o p -- python $V1 > $OUT/v1$P.log 2>&1; echo "v1 $P rc=$?"
  cp $(find $OUT/v1$P -name '*counter_collection.csv' | head -1) $OUT/v1_$P.csv 2>/dev/null; rm -rf $OUT/v1$P
done
python $R/scripts/pmc_round.py $OUT/pmc_hifigan_v1_x3_resblock.json "resblock_pair_x3_kernel" $OUT/v1_FETCH_SIZE.csv $OUT/v1_WRITE_SIZE.csv > $OUT/pmc_table_hifigan_v1.txt; head -30 $OUT/pmc_table_hifigan_v1.txt
rm -f $OUT/*.csv
