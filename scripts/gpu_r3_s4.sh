#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/s4; mkdir -p $OUT; cd $R
timeout 120 scripts/ubench/att_v2 2>&1 | tee $OUT/att_v2.txt | tail -25
for L in "" x3sall; do lib=tts_amd/libtts_amd${L:+_$L}.so; echo "== $lib"; TTSAMD_LIB_PATH=$lib timeout 200 python scripts/b1_latency.py 1 2>&1 | grep "two graphs\|all eager"; TTSAMD_LIB_PATH=$lib timeout 200 python bench.py --workload glow_hifigan_v2 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('configs[0] %.3f ms/sentence p50 %.2f' % (d['ms_per_step'], d['config']['sentence_latency_ms_p50']))"; done | tee $OUT/b1_x3sall.txt
TTSAMD_LIB_PATH=tts_amd/libtts_amd_x3sall.so timeout 400 python -m pytest tests/test_conv_gpu.py tests/test_hifigan_gpu.py tests/test_vits_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
(cd _old_r2 && for i in 1 2; do timeout 200 python -X faulthandler lanes_stress.py 400 2 5 > $OUT/old_lanes_$i.txt 2>&1; echo "old-code lanes stress (recapture every 5) $i rc=$? $(tail -1 $OUT/old_lanes_$i.txt | cut -c1-220)"; done)
for i in 1 2; do timeout 200 python -X faulthandler scripts/lanes_stress.py 400 2 5 > $OUT/new_lanes_$i.txt 2>&1; echo "new-code lanes stress (recapture every 5) $i rc=$? $(tail -1 $OUT/new_lanes_$i.txt | cut -c1-220)"; done
