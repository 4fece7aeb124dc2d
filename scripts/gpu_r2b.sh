#!/bin/bash
# Round-2 second GPU call: full tests, bench, phase clocks of the fused kernel, counter list, PMC passes on single kernels.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r2b; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
TTSAMD_LIB_PATH=$R/tts_amd/libtts_amd_clocks.so timeout 300 python scripts/resblock_phases.py 2>&1 | tee $OUT/phases.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1; grep -c . $OUT/counters.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
P3="GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 300 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -o p -- python $R/scripts/kernel_pmc_target.py > $OUT/pmc$i.log 2>&1; echo "pmc$i rc=$?"; tail -2 $OUT/pmc$i.log
  F=$(find $OUT/pmc$i -name '*counter_collection.csv' | head -1); [ -n "$F" ] && python $R/scripts/pmc_table.py $F > $OUT/pmc$i.txt && cat $OUT/pmc$i.txt
  rm -rf $OUT/pmc$i
done
