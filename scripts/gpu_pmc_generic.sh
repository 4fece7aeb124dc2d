#!/bin/bash
# usage: gpu_pmc_generic.sh <tag> "<counters>" <python args...>   -> per-kernel average of each counter
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=$1; CNT=$2; shift; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
PYTHONPATH=$R rocprofv3 --pmc $CNT --output-format csv -d $OUT/g -o g -- python "$@" > $OUT/g.log 2>&1
python - <<PY
import csv,collections
agg=collections.OrderedDict()
for r in csv.DictReader(open("$OUT/g/g_counter_collection.csv")):
    if 'ttsamd' in r['Kernel_Name']:
        k=(r['Kernel_Name'][8:60],r['Grid_Size'])
        d=agg.setdefault(k,collections.defaultdict(float)); d[r['Counter_Name']]+=float(r['Counter_Value']); d['n_'+r['Counter_Name']]+=1
for k,d in agg.items():
    n=max(v for kk,v in d.items() if kk.startswith('n_'))
    print(k, {kk:"%.4g"%(v/n) for kk,v in d.items() if not kk.startswith('n_')})
PY
