#!/bin/bash
# round 6: the product on the model-level C handles — same-box A/B of the headline step and the single request, Python host vs handle
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_native_ab; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_native_models_gpu.py tests/test_vits_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/pytest.txt
{
if [ -z "$SKIP_B32" ]; then for rep in 1 2; do for nat in 0 1; do
  echo "== B=32 step, TTSAMD_NATIVE_MODELS=$nat (pass $rep)"
  TTSAMD_NATIVE_MODELS=$nat timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-live-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %.2f ms/step  host=%s' % (d['ms_per_step'], d['config'].get('host')))"
done; done; fi
for rep in 1 2 3; do for nat in 0 1; do
  echo "== VITS B=1 request, TTSAMD_NATIVE_SINGLE=$nat (pass $rep)"
  TTSAMD_NATIVE_SINGLE=$nat timeout 300 python bench.py --workload vits_b1 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   p50 %.3f ms/request   %s' % (d['value'], {k: round(v, 3) for k, v in d.get('observed', {}).items() if isinstance(v, float)}))"
done; done
} 2>&1 | tee $OUT/native_ab.txt
