#!/bin/bash
# round 6: where do the fused-pair kernels' cycles go?  SQ counters of the C = 32 / 64 / 128 pairs and the small-K convs at the
# headline shapes (one counter set per rocprofv3 pass; quad-cycle units for SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*).
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_pmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 600 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -o p -- python $R/scripts/r6_pairs_ab.py ${PMC_WHAT:-pairs ups} > $OUT/p$i.log 2>&1; echo "pass $i rc=$?"
  cp $(find $OUT/p$i -name '*counter_collection.csv' | head -1) $OUT/p$i.csv 2>/dev/null; rm -rf $OUT/p$i
done
cd $R; python - <<'PY' | tee $OUT/pmc_pairs.txt
import csv, collections, glob, re
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in sorted(glob.glob("gpurun_out/r6_pmc/p*.csv")):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"].replace(" ", "")
        m = re.search(r"(resblock_pair_h2_kernel|conv1d_h2_kernel)<([^>]*)>", n)
        if not m: continue
        agg[m.group(1).replace("_kernel", "") + "<" + m.group(2) + ">"][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
print("per launch (sum over the chip, averaged over the launches); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles")
print("%-40s %4s %9s | of wave cycles: %8s %9s %7s | %8s %7s %8s %8s | %9s %9s %8s %8s %8s %9s" % (
    "kernel", "n", "cycles", "wait_any", "wait_inst", "active", "act_valu", "act_lds", "act_vmem", "wait_lds", "mfma_busy", "valu/mfma", "lds/mfma", "vmrd/mfma", "waves", "bank_conf"))
for k, d in agg.items():
    a = {c: sum(v.values()) / len(v) for c, v in d.items()}
    wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8.0 or 1
    mf = max(a.get("SQ_INSTS_MFMA", 1), 1)
    print("%-40s %4d %9.4g | %23.3f %9.3f %7.3f | %8.3f %7.3f %8.3f %8.3f | %9.3f %9.2f %8.2f %8.2f %8.3g %9.4f" % (
        k, len(d.get("SQ_WAVE_CYCLES", {})), cyc, a.get("SQ_WAIT_ANY", 0) / wc, a.get("SQ_WAIT_INST_ANY", 0) / wc, a.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        a.get("SQ_ACTIVE_INST_VALU", 0) / wc, a.get("SQ_ACTIVE_INST_LDS", 0) / wc, a.get("SQ_ACTIVE_INST_VMEM", 0) / wc, a.get("SQ_WAIT_INST_LDS", 0) / wc,
        a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / cyc, a.get("SQ_INSTS_VALU", 0) / mf, a.get("SQ_INSTS_LDS", 0) / mf, a.get("SQ_INSTS_VMEM_RD", 0) / mf,
        a.get("SQ_WAVES", 0), a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_LDS_IDX_ACTIVE", 1), 1)))

PY
rm -f $OUT/*.csv $OUT/*.log
