#!/bin/bash
# round 6: which 128- / 256-channel pairs to fuse, measured in the whole step (handles, default path): TTSAMD_FUSE_LIMITS A/B
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_pair256; mkdir -p $OUT; cd $R
for rep in 1 2; do for lim in "256:0" "256:3" "256:7" "256:11" "128:11,256:3"; do
  echo -n "B=32 TTSAMD_FUSE_LIMITS=$lim: "; TTSAMD_FUSE_LIMITS=$lim timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done | tee $OUT/limits_b32.txt
for rep in 1 2; do for lim in "256:0" "256:3" "256:7" "256:11" "128:11,256:11"; do
  echo -n "B=1 TTSAMD_FUSE_LIMITS=$lim: "; TTSAMD_FUSE_LIMITS=$lim timeout 600 python bench.py --workload vits_b1 --steps 100 --warmup 5 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['observed'].get('two_lanes_ms_per_request'))"
done; done | tee $OUT/limits_b1.txt
