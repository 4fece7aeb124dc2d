#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r2c; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -q -rf -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Extension" $OUT/pytest.log | tail -60
for F in 1 0; do
  TTSAMD_FUSE_RESBLOCKS=$F timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 8 > $OUT/bench_fuse$F.json 2> $OUT/bench_fuse$F.err; echo "bench fuse=$F rc=$?"
  python - <<PY
import json
d=json.load(open("$OUT/bench_fuse$F.json")); print("fuse=$F", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["all_conv_launches"])
PY
done
TTSAMD_FUSE_CHANNELS=32 timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 8 > $OUT/bench_fuse32.json 2>/dev/null
python -c "
import json; d=json.load(open('$OUT/bench_fuse32.json')); print('fuse32', d['ms_per_step'], d['roofline']['all_conv_launches'])"
