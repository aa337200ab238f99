#!/bin/bash
# Memory-side traffic + L2 hit rate of K5 in one bench regime: three rocprofv3 PMC passes (counters only) over
# `python3 tools/perf_k5.py pmc`, then tools/pmc_to_json.py (gfx950 FETCH_SIZE x2 correction, kernel-source sha stamp).
# usage: bash tools/pmc_traffic.sh <out-subdir-of-gpurun_out> <regime: r2|r1|locality|script> [fp8|pv]
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT
export RSA_PERF_REGIME=$2 RSA_PERF_NODENSE=1
KERN=bsfwd
if [ "$3" = "fp8" ]; then export RSA_PERF_FP8=1; KERN=bsfwd_fp8; fi
if [ "$3" = "pv" ]; then export RSA_PERF_FP8=pv; KERN=bsfwd_fp8; fi
cd /tmp; export TMPDIR=/tmp
N=0
for P in "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$((N+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/p$N -- python3 $R/tools/perf_k5.py pmc > $OUT/p$N.log 2>&1
done
cd $R
python3 tools/pmc_to_json.py $OUT/traffic.json $KERN "$OUT/**/*counter_collection.csv"
