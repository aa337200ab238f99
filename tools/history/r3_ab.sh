#!/bin/bash
# Round-3 A/B of K5 builds in ONE process on ONE device (+ PMC pair from the same lease):
#   r1 = round-1 head (adab149), r2 = round-2 head (c37b5bb), cur = the tree's library.
# usage (on the GPU box): bash tools/r3_ab.sh <tag>      outputs: gpurun_out/<tag>_ab.txt, gpurun_out/<tag>_pmc/
R=$PWD; TAG=${1:-r3a}; mkdir -p $R/gpurun_out
LIBS="r1=$R/build/librsa_hip_r1.so:14 r2=$R/build/librsa_hip_r2.so:18 r2nots=$R/build/librsa_hip_r2.so:18:nots cur=$R/rectified_spaattn_amd/librsa_hip.so:15 cur8=$R/rectified_spaattn_amd/librsa_hip.so:15:o8 curnots=$R/rectified_spaattn_amd/librsa_hip.so:15:nots"
python3 tools/ab_libs.py $LIBS --rounds ${ROUNDS:-12} > $R/gpurun_out/${TAG}_ab.txt 2>&1
tail -8 $R/gpurun_out/${TAG}_ab.txt
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_pmc; mkdir -p $OUT
N=0
for P in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" \
         "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  N=$((N+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/p$N -- python3 $R/tools/ab_libs.py r1=$R/build/librsa_hip_r1.so:14 r2=$R/build/librsa_hip_r2.so:18 cur=$R/rectified_spaattn_amd/librsa_hip.so:15 --pmc > /dev/null 2>&1
done
python3 $R/tools/pmc_summary.py "$OUT/**/*counter_collection.csv" > $OUT/summary.txt
cat $OUT/summary.txt
