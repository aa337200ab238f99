#!/bin/bash
# the two SQ counter passes of tools/pmc_passes.sh only (issue / wait / instruction mix), sparse call only
# usage: bash tools/pmc_sq.sh <out-subdir-of-gpurun_out>     (env RSA_K5_W64=1 RSA_TUNING=1: the 64-row K5)
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; export RSA_PERF_NODENSE=1
N=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES" \
         "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  N=$((N+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/p$N -- python3 $R/tools/perf_k5.py pmc > /dev/null 2>&1
done
python3 $R/tools/pmc_summary.py "$OUT/**/*counter_collection.csv" > $OUT/summary.txt
grep bsfwd $OUT/summary.txt
