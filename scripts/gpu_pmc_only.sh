#!/bin/bash
# the three PMC passes of the headline step alone (-> pmc_dominant_x3.json + table), stamped
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r03}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
HEAD=$(cat .build_head 2>/dev/null || echo unknown); STAMP=$(python -c "import bench; print(bench.code_stamp())" 2>/dev/null)
HDR="# commit $HEAD kernel-source stamp $STAMP ($(date -u +%Y-%m-%dT%H:%MZ), MI355X via gpurun)"
cd /tmp && export TMPDIR=/tmp
BENCH="$R/bench.py --serial-branches --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 600 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -o p -- python $BENCH > $OUT/pmc$i.log 2>&1; echo "pmc$i rc=$?"
  cp $(find $OUT/pmc$i -name '*counter_collection.csv' | head -1) $OUT/pmc$i.csv 2>/dev/null; rm -rf $OUT/pmc$i
done
(cd $R && python scripts/pmc_round.py $OUT/pmc_dominant_x3.json "conv1d_x3_kernel<11,1,1,4,4,1,0," $OUT/pmc1.csv $OUT/pmc2.csv $OUT/pmc3.csv > $OUT/pmc_table.txt); { echo "$HDR"; cat $OUT/pmc_table.txt; } > $OUT/pmc_table.txt.tmp && mv $OUT/pmc_table.txt.tmp $OUT/pmc_table.txt; head -8 $OUT/pmc_table.txt; cat $OUT/pmc_dominant_x3.json
V1="$R/bench.py --workload hifigan_v1 --items 16 --steps 1 --warmup 1 --no-cpu-baseline"
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  PYTHONPATH=$R timeout 600 rocprofv3 --pmc $P --output-format csv -d $OUT/v1$P -o p -- python $V1 > $OUT/v1$P.log 2>&1; echo "v1 $P rc=$?"
  cp $(find $OUT/v1$P -name '*counter_collection.csv' | head -1) $OUT/v1_$P.csv 2>/dev/null; rm -rf $OUT/v1$P
done
(cd $R && python scripts/pmc_round.py $OUT/pmc_hifigan_v1_x3_resblock.json "resblock_pair_x3_kernel" $OUT/v1_FETCH_SIZE.csv $OUT/v1_WRITE_SIZE.csv > $OUT/pmc_table_hifigan_v1.txt); { echo "$HDR"; cat $OUT/pmc_table_hifigan_v1.txt; } > $OUT/pmc_table_hifigan_v1.txt.tmp && mv $OUT/pmc_table_hifigan_v1.txt.tmp $OUT/pmc_table_hifigan_v1.txt
rm -f $OUT/*.csv
