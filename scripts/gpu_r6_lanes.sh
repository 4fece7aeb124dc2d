#!/bin/bash
# round 6: request lanes x branch streams of the B = 32 step, same box
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_lanes; mkdir -p $OUT; cd $R
for rep in 1 2; do for l in 1 2 3 4; do for sb in "" "--serial-branches"; do
  echo -n "lanes=$l $sb: "; timeout 600 python bench.py --steps 12 --warmup 3 --lanes $l $sb --no-extras --no-cpu-baseline --no-live-pmc 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done; done | tee $OUT/lanes.txt
