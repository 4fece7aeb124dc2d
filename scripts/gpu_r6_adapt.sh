#!/bin/bash
# round 6: tile width chosen by padded columns (TTSAMD_H2_ADAPT_TILES): conv parity, then the B = 32 step with / without, same box
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_adapt; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_vits_gpu.py tests/test_text_gpu.py tests/test_glow_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/pytest.txt
for rep in 1 2 3; do for v in 0 1; do echo -n "B=32 TTSAMD_H2_ADAPT_TILES=$v: "; TTSAMD_H2_ADAPT_TILES=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done; done | tee $OUT/bench_ab.txt
