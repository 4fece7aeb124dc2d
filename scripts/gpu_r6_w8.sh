#!/bin/bash
# round 6: eight-wave blocks (TTSAMD_H2_W8_MAX) for launches that do not fill the chip once: a single VITS request; parity under the switch
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_w8b; mkdir -p $OUT; cd $R
TTSAMD_H2_W8_MAX=600 timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_hifigan_gpu.py tests/test_vits_gpu.py tests/test_native_models_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/pytest.txt
for rep in 1 2 3; do for v in 0 128 600 1100; do
  echo -n "B=1 TTSAMD_H2_W8_MAX=$v: "; TTSAMD_H2_W8_MAX=$v timeout 300 python scripts/b1_quick.py 60 2>&1 | grep "p50"
done; done | tee $OUT/b1.txt
for v in 0 600; do echo -n "configs[0] W8_MAX=$v: "; TTSAMD_H2_W8_MAX=$v timeout 300 python bench.py --workload glow_hifigan_v2 --no-cpu-baseline --no-live-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done | tee -a $OUT/b1.txt
for v in 0 600; do echo -n "B=32 W8_MAX=$v: "; TTSAMD_H2_W8_MAX=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done | tee -a $OUT/b1.txt
