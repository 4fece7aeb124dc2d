#!/bin/bash
# confidence soak on the final code: the GPU suite three times, the lanes stress 2 x 1000 rounds, the default bench three times
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/soak; mkdir -p $OUT; cd $R
for i in 1 2 3; do timeout 600 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider -o faulthandler_timeout=250 > $OUT/pytest_$i.log 2>&1; echo "suite $i rc=$? $(tail -1 $OUT/pytest_$i.log)"; done
for i in 1 2; do timeout 300 python -X faulthandler scripts/lanes_stress.py 1000 2 5 2>&1 | tail -1; done
for i in 1 2 3; do S=$(date +%s); python bench.py --steps 20 --warmup 5 2>/dev/null > $OUT/bench_$i.jsonl; python -c "
import json
for ln in open('$OUT/bench_$i.jsonl'):
    d=json.loads(ln)
    if 'extra_workload' in d: print('  ', d['extra_workload'], '%.3f ms/step' % d['line']['ms_per_step'])
    elif 'metric' in d: print('bench $i: %.2f ms/step value %.4g frac %.4f  wall $(( $(date +%s) - S )) s, last line %d bytes' % (d['ms_per_step'], d['value'], d['roofline']['frac'], len(ln)))
"; done
