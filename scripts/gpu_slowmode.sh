#!/bin/bash
# Is this box in the "slow mode" of the single-sentence loop (1.47 vs 1.85 ms per sentence), and what moves it?
# One fresh process as found; if it is slow (or FORCE=1): the DPM state files, two processes at `rocm-smi --setperflevel high`,
# back to auto, one more as found, and a kernel trace (per-kernel medians: compare with the fast-mode table of
# profiles/r05_bimodal_probe.txt).      bash scripts/gpu_slowmode.sh [outdir]
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-slowmode}; mkdir -p $OUT; cd $R
dpm() { for f in /sys/class/drm/card*/device/power_dpm_force_performance_level; do [ -e $f ] && echo "$(dirname $f | xargs dirname | xargs basename): perf level $(cat $f), busy $(cat $(dirname $f)/gpu_busy_percent 2>/dev/null)%, sclk $(grep "\*" $(dirname $f)/pp_dpm_sclk | tr -d "\n")"; done; }
{
echo "--- as found"; dpm
timeout 120 python scripts/bimodal_step.py 300 asfound1 2>&1 | grep BIMODAL
} > $OUT/slowmode.txt 2>&1
MS=$(grep asfound1 $OUT/slowmode.txt | sed -E 's/.*ms\/sentence ([0-9.]+).*/\1/')
SLOW=$(python -c "print(1 if float('${MS:-0}') > 1.65 else 0)")
echo "as found: $MS ms/sentence (slow=$SLOW)"
if [ "$SLOW" = "1" ] || [ -n "$FORCE" ]; then
{
timeout 120 python scripts/bimodal_step.py 300 asfound2 2>&1 | grep BIMODAL
echo "--- rocm-smi --setperflevel high"; rocm-smi --setperflevel high 2>&1 | grep -v "^=\|^$" | head -3; dpm
for i in 1 2; do timeout 120 python scripts/bimodal_step.py 300 perfhigh$i 2>&1 | grep BIMODAL; done
echo "--- rocm-smi --setperflevel auto"; rocm-smi --setperflevel auto 2>&1 | grep -v "^=\|^$" | head -3
timeout 120 python scripts/bimodal_step.py 300 auto_again 2>&1 | grep BIMODAL
echo "--- after 6 s of a chip-filling kernel (the headline's conv) in another process, then the sentence loop"
timeout 60 python -c "
import sys, time, torch
sys.path.insert(0, '.')
from tts_amd import ops
x = torch.randn(32, 128, 49280, device='cuda:0'); y = torch.empty_like(x)
pc = ops.PackedConv(torch.randn(128, 128, 11) / 37.5, None, 'cuda:0')
t0 = time.time()
while time.time() - t0 < 6:
    for _ in range(50): ops.conv1d(pc, x, y, in_act=ops.ACT_LRELU, in_slope=0.1, res=x)
    torch.cuda.synchronize()
" 2>&1 | grep -v amdgpu.ids
timeout 120 python scripts/bimodal_step.py 300 after_load 2>&1 | grep BIMODAL
} >> $OUT/slowmode.txt 2>&1
cd /tmp; export TMPDIR=/tmp
PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o g -- python $R/scripts/bimodal_step.py 300 traced_slow > $OUT/tr.log 2>&1
grep BIMODAL $OUT/tr.log >> $OUT/slowmode.txt
python $R/scripts/bimodal_trace.py $(find $OUT/tr -name '*kernel_trace.csv' | head -1) traced_slow >> $OUT/slowmode.txt 2>&1
rm -rf $OUT/tr
fi
cat $OUT/slowmode.txt
