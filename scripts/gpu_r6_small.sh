#!/bin/bash
# round 6: branch-free loads in the text-length kernels (channel_norm_small, spline, convflow_pre): parity + the request's timeline
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_small; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_text_gpu.py tests/test_vits_gpu.py tests/test_glow_gpu.py tests/test_native_models_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/pytest.txt
bash scripts/gpu_b1_tl.sh r6_small/tl > /dev/null 2>&1
python - <<'PY' | tee $OUT/summary.txt
import re, collections
rows = collections.defaultdict(list)
for l in open("gpurun_out/r6_small/tl/b1_timeline.txt"):
    m = re.match(r"\s*[\d.]+ \+\s+([\d.]+) us\s+q\d+ gap\s+[\d.]+\s+blocks\s+(\d+)\s+(.*)", l)
    if m: rows[m.group(3).strip()[:44]].append(float(m.group(1)))
print(open("gpurun_out/r6_small/tl/b1_timeline.txt").readline().strip())
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    if "h2_kernel" in k or "resblock" in k: continue
    print("%-46s n=%3d  sum %7.1f us  median %5.1f" % (k, len(v), sum(v), sorted(v)[len(v)//2]))
PY
for rep in 1 2 3; do timeout 600 python bench.py --workload vits_b1 --steps 100 --warmup 5 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vits_b1', d['value'], d['observed'].get('two_lanes_ms_per_request'))"; done | tee -a $OUT/summary.txt
timeout 600 python bench.py --workload glow_hifigan_v2 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('configs[0]', d['value'], d['unit'], d.get('ms_per_step'))" | tee -a $OUT/summary.txt
