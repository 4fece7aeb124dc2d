#!/bin/bash
# effective shader clock of a kernel: GRBM_GUI_ACTIVE / duration, one PMC pass.  usage: gpu_clock.sh <tag> <python args>
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=$1; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
PYTHONPATH=$R rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/clk -o clk -- python "$@" > $OUT/clk.log 2>&1
python - <<PY
import csv,collections
agg=collections.defaultdict(dict)
for r in csv.DictReader(open("$OUT/clk/clk_counter_collection.csv")):
    if 'conv1d' in r['Kernel_Name']:
        d=agg[r['Dispatch_Id']]; d[r['Counter_Name']]=float(r['Counter_Value']); d['us']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3; d['grid']=r['Grid_Size']
for k,d in list(agg.items())[-6:]:
    print(k,d, "GHz(GUI_ACTIVE/dur)=%.3f"%(d.get('GRBM_GUI_ACTIVE',0)/d['us']/1e3), "mfma_busy/1024/gui=%.3f"%(d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/max(d.get('GRBM_GUI_ACTIVE',1),1)))
PY
