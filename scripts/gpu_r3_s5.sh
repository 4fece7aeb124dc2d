#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/s5; mkdir -p $OUT; cd $R
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x -rf -o faulthandler_timeout=250 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Extension" $OUT/pytest.log | tail -12
timeout 300 python bench.py --workload glow_hifigan_v2 --steps 50 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_glow.json; python -c "import json; d=json.load(open('$OUT/bench_glow.json')); print('configs[0] %.3f ms/sentence p50 %.2f  cpu %s' % (d['ms_per_step'], d['config']['sentence_latency_ms_p50'], d.get('cpu_baseline', {}).get('sample')))"
timeout 200 python scripts/b1_latency.py 1 2>&1 | grep "two graphs\|all eager"
