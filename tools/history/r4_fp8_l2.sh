#!/bin/bash
# L2 hit rate of the e4m3 / 32-row K5 with aligned starts off / on (k5_gsync bit 1)
export RSA_TUNING=1
R=$PWD; L=rectified_spaattn_amd/librsa_hip.so
cd /tmp; export TMPDIR=/tmp
for G in 0 3; do
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/r4n_l2fp8_$G -- python3 $R/tools/ab_libs.py x=$R/$L::k5_gsync=$G --pmc --fp8 > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/r4n_l232_$G -- python3 $R/tools/ab_libs.py x=$R/$L::k5_w64=0,k5_gsync=$G --pmc > /dev/null 2>&1
done
cd $R
python3 tools/l2_hits.py gpurun_out/r4n_l2fp8_0 gpurun_out/r4n_l2fp8_3 gpurun_out/r4n_l232_0 gpurun_out/r4n_l232_3
