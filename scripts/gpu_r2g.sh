#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r2g; mkdir -p $OUT; cd $R
for D in 0 1 2; do
  TTSAMD_MAS_DEBUG=$D timeout 300 python bench.py --workload mas --steps 30 --no-cpu-baseline > $OUT/mas_dbg$D.json 2>>$OUT/err.log
  python -c "
import json
d=json.load(open('$OUT/mas_dbg$D.json')); print('dbg=$D', '%.3f ms'%d['ms_per_step'])"
done
