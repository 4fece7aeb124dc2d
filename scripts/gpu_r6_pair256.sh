#!/bin/bash
# round 6: 256-channel k = 3 pairs fused in both hosts: parity suites touching the vocoder, then same-box A/B (Python host: env switch)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_pair256; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_resblock_gpu.py tests/test_hifigan_gpu.py tests/test_native_models_gpu.py tests/test_vits_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/pytest.txt
for rep in 1 2; do for fc in "8,16,32,64,128" "8,16,32,64,128,256"; do
  echo -n "python host, fuse_channels=$fc: "; TTSAMD_NATIVE_MODELS=0 TTSAMD_FUSE_CHANNELS=$fc timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done | tee $OUT/bench_ab.txt
echo -n "handles (default): "; timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" | tee -a $OUT/bench_ab.txt
for rep in 1 2; do for fc in "8,16,32,64,128" "8,16,32,64,128,256"; do
  echo -n "B=1 python host, fuse_channels=$fc: "; TTSAMD_NATIVE_MODELS=0 TTSAMD_FUSE_CHANNELS=$fc timeout 600 python bench.py --workload vits_b1 --steps 100 --warmup 5 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
done; done | tee $OUT/b1_ab.txt
