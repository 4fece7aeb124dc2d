#!/bin/bash
# stall breakdown of conv launches.  usage: gpu_pmc_conv.sh <tag> <conv_micro specs...>
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=$1; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
PYTHONPATH=$R rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/st -o st -- python $R/scripts/conv_micro.py "$@" > $OUT/st.log 2>&1
PYTHONPATH=$R rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/in -o in -- python $R/scripts/conv_micro.py "$@" > $OUT/in.log 2>&1
python - <<PY
import csv,collections
for f in ("st/st","in/in"):
    agg=collections.OrderedDict()
    for r in csv.DictReader(open("$OUT/%s_counter_collection.csv"%f)):
        if 'conv1d' in r['Kernel_Name']:
            k=(r['Kernel_Name'][26:50],r['Grid_Size'])
            d=agg.setdefault(k,collections.defaultdict(float)); d[r['Counter_Name']]+=float(r['Counter_Value']); d['n_'+r['Counter_Name']]+=1
    for k,d in agg.items():
        n=max(v for kk,v in d.items() if kk.startswith('n_'))
        print(k, {kk:"%.4g"%(v/n) for kk,v in d.items() if not kk.startswith('n_')})
PY
