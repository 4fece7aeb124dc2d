#!/bin/bash
# MAS: the branch-free column step against the round-3 one (same box), bit-exactness suite on both.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_mas; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_mas_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/pytest.txt
rm -f $OUT/mas_ab.txt
for pass in 1 2; do
  for V in 1 0; do
    echo "== TTSAMD_MAS_MW=$V TTSAMD_MAS_BT=$V (pass $pass; 1 = round-3 column step + round-2 backtrack walk, 0 = round-6 kernels)" | tee -a $OUT/mas_ab.txt
    TTSAMD_MAS_MW=$V TTSAMD_MAS_BT=$V timeout 300 python scripts/r6_mas_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/mas_ab.txt
  done
done
cd /tmp && export TMPDIR=/tmp
for V in 1 0; do
  TTSAMD_MAS_MW=$V TTSAMD_MAS_BT=$V PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$V -o t -- python $R/bench.py --workload mas --mas-batch 32 --steps 50 --no-cpu-baseline > $OUT/prof$V.log 2>&1
  S=$(find $OUT/prof$V -name '*kernel_stats.csv' | head -1); echo "== kernel stats TTSAMD_MAS_MW=$V" | tee -a $OUT/mas_ab.txt; head -6 $S | cut -c1-200 | tee -a $OUT/mas_ab.txt; rm -rf $OUT/prof$V $OUT/prof$V.log
done
