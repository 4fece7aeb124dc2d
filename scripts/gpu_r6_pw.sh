#!/bin/bash
# round 6: the fused [norm ->] 1x1 conv -> norm launch (ttsamd_pw_norm): tests, then the single-request latencies with it off / on
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_pw; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_text_gpu.py tests/test_native_models_gpu.py tests/test_vits_gpu.py tests/test_glow_gpu.py tests/test_synth_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/pytest.txt
rm -f $OUT/ab.txt
for pass in 1 2; do for V in 0 1; do
  echo "== TTSAMD_PW_NORM=$V (pass $pass)" | tee -a $OUT/ab.txt
  TTSAMD_PW_NORM=$V timeout 300 python bench.py --workload vits_b1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  vits_b1  p50 %.3f ms  two lanes %.3f  launches %s' % (d['value'], d['observed']['two_lanes_ms_per_request'], d['observed']['launches_per_request']))" | tee -a $OUT/ab.txt
  TTSAMD_PW_NORM=$V timeout 300 python bench.py --workload glow_hifigan_v2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  configs[0] %.3f ms/sentence  launches %s' % (d['ms_per_step'], d['roofline']['launches_per_request']))" | tee -a $OUT/ab.txt
done; done
