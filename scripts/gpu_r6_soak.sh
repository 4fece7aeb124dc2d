#!/bin/bash
# full GPU suite several times on one box (fresh process each; the first run is the box's first touch of every kernel): any
# failure or abort keeps its whole log
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_soak; mkdir -p $OUT; cd $R
for i in ${SOAK_RUNS:-1 2 3 4 5}; do
  timeout 900 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/run$i.log 2>&1; rc=$?
  echo "run $i rc=$rc $(grep -v '^  File\|^Extension' $OUT/run$i.log | tail -1)" | tee -a $OUT/soak.txt
  [ $rc -eq 0 ] && rm -f $OUT/run$i.log || { tail -c 200000 $OUT/run$i.log > $OUT/run$i.tail; rm -f $OUT/run$i.log; }
done
