#!/bin/bash
# rocprofv3 PMC passes (counters only) over the mask-selection pass K1..K4: `python3 tools/perf_k5.py pmcsel`.
# usage: bash tools/pmc_select.sh <out-subdir-of-gpurun_out>
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
N=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES" \
         "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  N=$((N+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/p$N -- python3 $R/tools/perf_k5.py pmcsel > /dev/null 2>&1
done
python3 $R/tools/pmc_summary.py "$OUT/**/*counter_collection.csv" > $OUT/summary.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
cat $OUT/summary.txt
