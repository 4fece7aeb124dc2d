#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/x3wpmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for P in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 300 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -o p -- python $R/scripts/ubench/x3w_pmc_target.py > $OUT/p$i.log 2>&1; echo "pmc$i rc=$?"
  F=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  python - "$F" <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    if "x3w" in nm or "conv1d_x3_kernel" in nm:
        agg[nm[:50]][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
for k, v in agg.items():
    print(k)
    for c, lst in v.items():
        per = collections.defaultdict(float)
        for d, val in lst: per[d] += val
        print("    %-28s %.4g" % (c, sum(per.values()) / len(per)))
PY
  rm -rf $OUT/p$i
done
