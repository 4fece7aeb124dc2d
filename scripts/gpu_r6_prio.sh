#!/bin/bash
# round 6: MRF branch streams with queue priorities (largest kernel size high, the others low) for a single request
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_prio; mkdir -p $OUT; cd $R
for rep in 1 2; do for p in 0 1; do
  echo "== TTSAMD_BRANCH_PRIORITY=$p"; TTSAMD_BRANCH_PRIORITY=$p timeout 300 python scripts/b1_latency.py 1 2>&1 | grep -v amdgpu.ids | grep "B=1"
done; done | tee $OUT/prio.txt
