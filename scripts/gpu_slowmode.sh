#!/bin/bash
# Which regime of the single-sentence loop is this box in (1.47 vs 1.85 ms per sentence), and what distinguishes
# the boxes?  Box configuration (partition modes, xnack, firmware, driver), dependent-load latencies of the memory paths
# (scripts/ubench/latency_probe.hip), one fresh process of the loop with our card's clocks and the neighbours' load, and a kernel
# trace with per-kernel medians — the fast and the slow table differ in a handful of kernels, not uniformly.
#   bash scripts/gpu_slowmode.sh [outdir]
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-slowmode}; mkdir -p $OUT; cd $R
{
echo "--- box: $(hostname) kernel $(uname -r)"
rocminfo 2>/dev/null | grep -i -m3 "gfx950\|xnack" | tr -s ' '
echo "amdgpu noretry=$(cat /sys/module/amdgpu/parameters/noretry 2>/dev/null) vm_fragment_size=$(cat /sys/module/amdgpu/parameters/vm_fragment_size 2>/dev/null) iommu: $(grep -o 'iommu=[a-z]*\|amd_iommu=[a-z]*' /proc/cmdline | tr '\n' ' ')"
rocm-smi --showcomputepartition --showmemorypartition --showvbios --showdriverversion 2>/dev/null | grep -v "^=\|^$\|WARNING" | head -24
echo "HSA_XNACK=${HSA_XNACK:-unset} HIP_VISIBLE_DEVICES=${HIP_VISIBLE_DEVICES:-unset} ROCR_VISIBLE_DEVICES=${ROCR_VISIBLE_DEVICES:-unset}"
echo "--- dependent-load latencies"; ./scripts/ubench/latency_probe 2>&1
echo "--- sentence loop, one fresh process"
timeout 120 python scripts/bimodal_step.py 300 asfound 2>&1 | grep BIMODAL
} > $OUT/slowmode.txt 2>&1
cd /tmp; export TMPDIR=/tmp
PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o g -- python $R/scripts/bimodal_step.py 300 traced > $OUT/tr.log 2>&1
grep BIMODAL $OUT/tr.log >> $OUT/slowmode.txt
python $R/scripts/bimodal_trace.py $(find $OUT/tr -name '*kernel_trace.csv' | head -1) traced >> $OUT/slowmode.txt 2>&1
rm -rf $OUT/tr $OUT/tr.log
MS=$(grep "BIMODAL asfound" $OUT/slowmode.txt | sed -E 's/.*ms\/sentence ([0-9.]+).*/\1/')
if [ "$(python -c "print(1 if float('${MS:-0}') > 1.65 else 0)")" = "1" ] || [ -n "$FORCE_VARIANTS" ]; then
  # a slow-regime box: the two sensitive kernels in their alternative forms (attention: the 8-wave v2 kernel forced; flow-block tail:
  # affine-coupling conv + separate InvConv / ActNorm kernel instead of the fused MIX epilogue), each traced
  for V in "TTSAMD_ATT_V2=1" "TTSAMD_ATT_V2=0" "GLOW_FUSE_MIX=0"; do
    env $V PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trv -o g -- python $R/scripts/bimodal_step.py 300 "$V" > $OUT/trv.log 2>&1
    grep BIMODAL $OUT/trv.log | cut -c1-140 >> $OUT/slowmode.txt
    python $R/scripts/bimodal_trace.py $(find $OUT/trv -name '*kernel_trace.csv' | head -1) "$V" 2>&1 | grep "TRACE\|attention\|1, 1, 7, 4\|1, 1, 5, 4\|invconv" >> $OUT/slowmode.txt
    rm -rf $OUT/trv $OUT/trv.log
  done
fi
cat $OUT/slowmode.txt
