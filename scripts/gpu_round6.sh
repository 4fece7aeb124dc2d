#!/bin/bash
# Round-6 evidence in one GPU call, every file stamped with the commit and the kernel-source stamp it was recorded with: box state
# and single-sentence regime, GPU tests, smoke, the default bench run (details + four extra lines + headline), MAS, B=1 latency,
# per-kernel timeline of the single-sentence chain, rocprofv3 kernel stats + per-shape medians of the headline step, PMC passes
# (traffic, matrix-pipe busy, cycles) with the per-instantiation traffic table of the fused pairs.
#   git rev-parse --short HEAD > .build_head; gpurun --timeout 2400 -- 'bash scripts/gpu_round6.sh r06'
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r06}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
HEAD=$(cat .build_head 2>/dev/null || echo unknown); STAMP=$(python -c "import bench; print(bench.code_stamp())" 2>/dev/null)
HDR="# commit $HEAD kernel-source stamp $STAMP ($(date -u +%Y-%m-%dT%H:%MZ), MI355X via gpurun)"
stamp() { f=$1; { echo "$HDR"; cat $f; } > $f.tmp && mv $f.tmp $f; }
bash scripts/gpu_slowmode.sh $TAG/slow > /dev/null 2>&1; cp $OUT/slow/slowmode.txt $OUT/box_and_sentence_regime.txt; stamp $OUT/box_and_sentence_regime.txt; grep "BIMODAL\|HBM \|L2 " $OUT/box_and_sentence_regime.txt | cut -c1-200; rm -rf $OUT/slow
timeout 1200 python -m pytest tests -m gpu -q -rf -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Extension" $OUT/pytest.log | tail -3 > $OUT/gpu_tests.txt; stamp $OUT/gpu_tests.txt; tail -2 $OUT/gpu_tests.txt
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log | tee -a $OUT/gpu_tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_n1.jsonl 2> $OUT/bench_n1.err; echo "bench rc=$?"; tail -5 $OUT/bench_n1.jsonl | cut -c1-300; awk '{print length($0)}' $OUT/bench_n1.jsonl | tr '\n' ' '; echo
for B in 32 256; do timeout 300 python bench.py --workload mas --mas-batch $B --steps 20 2>/dev/null | tail -1 > $OUT/bench_mas_b$B.json; cut -c1-200 $OUT/bench_mas_b$B.json; done
{ timeout 300 python scripts/b1_latency.py 1 2>&1 | grep -v amdgpu.ids; timeout 200 python scripts/b1_quick.py 2>&1 | grep -v amdgpu.ids; } > $OUT/b1_latency.txt; stamp $OUT/b1_latency.txt; cat $OUT/b1_latency.txt
bash scripts/gpu_b1_tl.sh $TAG/tlb1 > /dev/null 2>&1; { echo "$HDR"; cat $OUT/tlb1/b1_timeline.txt; } > $OUT/b1_timeline.txt; head -2 $OUT/b1_timeline.txt; rm -rf $OUT/tlb1
{ timeout 400 python scripts/att_ab.py 2>&1 | grep -v amdgpu.ids; } > $OUT/attention_v3_ab.txt; stamp $OUT/attention_v3_ab.txt; grep "T=257" $OUT/attention_v3_ab.txt
timeout 400 python scripts/r6_pairs_ab.py pairs convs ups 2>&1 | grep -v amdgpu.ids > $OUT/kernels_at_headline_shapes.txt; stamp $OUT/kernels_at_headline_shapes.txt; head -4 $OUT/kernels_at_headline_shapes.txt
cd /tmp && export TMPDIR=/tmp
BENCH="$R/bench.py --serial-branches --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-live-pmc"
PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $R/bench.py --serial-branches --lanes 1 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-live-pmc > $OUT/prof.log 2>&1; echo "prof rc=$?"
S=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); T=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
python $R/scripts/prof_summary.py stats $S > $OUT/kernel_stats.txt; stamp $OUT/kernel_stats.txt; head -8 $OUT/kernel_stats.txt | cut -c1-160
python $R/scripts/trace_shapes.py $T 16 200 > $OUT/per_shape.txt; stamp $OUT/per_shape.txt; head -4 $OUT/per_shape.txt
rm -rf $OUT/prof
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 600 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -o p -- python $BENCH > $OUT/pmc$i.log 2>&1; echo "pmc$i rc=$?"
  cp $(find $OUT/pmc$i -name '*counter_collection.csv' | head -1) $OUT/pmc$i.csv 2>/dev/null; rm -rf $OUT/pmc$i
done
(cd $R && python scripts/pmc_round.py $OUT/pmc_dominant_h2.json "conv1d_h2_kernel<11,1,1,4,4,1,0>" $OUT/pmc1.csv $OUT/pmc2.csv $OUT/pmc3.csv > /dev/null); cat $OUT/pmc_dominant_h2.json | cut -c1-400
(cd $R && python scripts/pmc_table_full.py $OUT/pmc1.csv $OUT/pmc2.csv $OUT/pmc3.csv $OUT/pmc4.csv > $OUT/pmc_table.txt); stamp $OUT/pmc_table.txt; head -30 $OUT/pmc_table.txt | cut -c1-200
(cd $R && python scripts/pair_traffic_table.py $OUT/pmc1.csv $OUT/pmc2.csv > $OUT/pair_traffic.txt); stamp $OUT/pair_traffic.txt; tail -4 $OUT/pair_traffic.txt
rm -f $OUT/*.csv $OUT/pmc*.log $OUT/prof.log
