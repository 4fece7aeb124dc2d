#!/bin/bash
# configs[0] bimodality hunt: N fresh processes of the sentence loop with clocks sampled, the same under knobs,
# then under rocprofv3 --kernel-trace with per-sentence kernel-time / idle-time summaries.   bash scripts/gpu_bimodal.sh [outdir]
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-bimodal}; mkdir -p $OUT; cd $R
{
rocm-smi --showperflevel --showclocks 2>&1 | grep -v "^=\|^$" | head -12
for i in 1 2 3 4 5 6 7 8; do timeout 120 python scripts/bimodal_step.py 300 plain$i 2>&1 | grep BIMODAL; done
for i in 1 2 3; do GPU_MAX_HW_QUEUES=1 timeout 120 python scripts/bimodal_step.py 300 hwq1_$i 2>&1 | grep BIMODAL; done
for i in 1 2 3; do HSA_ENABLE_SDMA=0 timeout 120 python scripts/bimodal_step.py 300 nosdma_$i 2>&1 | grep BIMODAL; done
for i in 1 2 3; do HIP_LAUNCH_BLOCKING=0 AMD_SERIALIZE_KERNEL=0 GPU_MAX_HW_QUEUES=2 timeout 120 python scripts/bimodal_step.py 300 hwq2_$i 2>&1 | grep BIMODAL; done
echo "--- perf level high (rocm-smi --setperflevel high) ---"
rocm-smi --setperflevel high 2>&1 | grep -v "^=\|^$" | head -3
for i in 1 2 3 4; do timeout 120 python scripts/bimodal_step.py 300 perfhigh$i 2>&1 | grep BIMODAL; done
rocm-smi --setperflevel auto 2>&1 | grep -v "^=\|^$" | head -3
} > $OUT/bimodal_runs.txt 2>&1
cat $OUT/bimodal_runs.txt
cd /tmp; export TMPDIR=/tmp
for i in 1 2 3 4 5; do
  PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr$i -o g -- python $R/scripts/bimodal_step.py 300 traced$i > $OUT/tr$i.log 2>&1
  grep BIMODAL $OUT/tr$i.log
  T=$(find $OUT/tr$i -name '*kernel_trace.csv' | head -1)
  python $R/scripts/bimodal_trace.py $T traced$i > $OUT/trace$i.txt 2>&1; head -1 $OUT/trace$i.txt
  rm -rf $OUT/tr$i
done
