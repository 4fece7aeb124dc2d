#!/bin/bash
# PMC HBM traffic of the dominant conv kernel (two passes) -> gpurun_out/<tag>/pmc_dominant_<prec>.json
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-pmc}; PREC=${2:-x3}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  PYTHONPATH=$R timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o pmc -- python $R/bench.py --precision $PREC --serial-branches --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$C.log 2>&1
  echo "$C rc=$?"
done
F=$(find $OUT/FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find $OUT/WRITE_SIZE -name '*counter_collection.csv' | head -1)
SUB="conv1d_mfma_kernel<11,1,2,2,2,2,0>"; [ "$PREC" = "x3" ] && SUB="conv1d_x3_kernel<11,1,1,4,4,1,0>"
python $R/scripts/pmc_dominant.py $F $W "$SUB" $OUT/pmc_dominant_$PREC.json
python $R/scripts/prof_summary.py pmc $F > $OUT/pmc_fetch.txt; python $R/scripts/prof_summary.py pmc $W > $OUT/pmc_write.txt
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
