#!/bin/bash
# L2 hit rate of K5 when only the FIRST generation of workgroups runs (all start together): heavy 48 + 512 sparse
set -x
export RSA_TUNING=1
for RG in locality r2; do
  for MB in 560 1072; do
    OUT=gpurun_out/r2e_${RG}_${MB}; mkdir -p $OUT
    ( cd /tmp; export TMPDIR=/tmp; RSA_K5_MAXBLOCKS=$MB RSA_PERF_REGIME=$RG RSA_PERF_NODENSE=1 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d /root/repo/$OUT -- python3 /root/repo/tools/perf_k5.py pmc > /root/repo/$OUT.log 2>&1 )
    python3 tools/pmc_summary.py "$OUT/**/*counter_collection.csv" | grep bsfwd
  done
done
