#!/bin/bash
export RSA_TUNING=1
mkdir -p gpurun_out
T=${1:-r4l}
L=rectified_spaattn_amd/librsa_hip.so
for H in 24 3; do
  ( RSA_PERF_H=$H timeout 600 python tools/ab_libs.py first=$L::k5_w64=1,k5_gsync=1,k5_text_last=0 last=$L::k5_w64=1,k5_gsync=1,k5_text_last=1 free=$L::k5_w64=1,k5_gsync=0,k5_text_last=0 --rounds 8 ) > gpurun_out/${T}_ab_h$H.txt 2>&1
  echo heads $H; tail -3 gpurun_out/${T}_ab_h$H.txt | cut -c1-200
done
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gsync.py tests/test_gpu_shard_invariance.py -x -q -m gpu 2>&1 | tail -2 )
