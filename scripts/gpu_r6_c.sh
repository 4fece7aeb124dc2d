R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_c; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_vits_gpu.py tests/test_native_models_gpu.py -m gpu -q -x -p no:cacheprovider -s -k "trained_like or native" 2>&1 | grep -v amdgpu.ids | grep "trained-like\|passed\|failed\|Error\|assert" | tee $OUT/pytest.txt
for rep in 1 2; do for v in 0 2; do PAIR_VARIANT=$v timeout 300 python scripts/r6_pairs_ab.py pairs 2>&1 | grep -v amdgpu.ids | grep "C=64\|C=32"; done; done | tee $OUT/pairs_variant_ab.txt
SKIP_B32=1 bash scripts/gpu_r6_native_ab.sh 2>&1 | grep -v "^\.\|passed" | tee $OUT/b1_ab.txt
