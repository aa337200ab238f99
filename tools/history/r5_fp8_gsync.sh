#!/bin/bash
export RSA_TUNING=1
L=rectified_spaattn_amd/librsa_hip.so
for WL in wan22_ti2v_720p_121f wan21_720p_81f; do
  RSA_PERF_WORKLOAD=$WL RSA_PERF_REGIME=r2 python tools/ab_libs.py free=$L::k5_gsync=1 aligned=$L::k5_gsync=3 --rounds 8 --fp8 2>&1 | grep -v amdgpu | tail -2 | sed "s/^/$WL fp8 r2: /"
done
RSA_PERF_REGIME=script python tools/ab_libs.py free=$L::k5_gsync=1 aligned=$L::k5_gsync=3 --rounds 6 --fp8 2>&1 | grep -v amdgpu | tail -2 | sed "s/^/hunyuan fp8 script: /"
RSA_PERF_WORKLOAD=wan22_ti2v_720p_121f RSA_PERF_REGIME=r2 python tools/ab_libs.py r5=$L::k5_gsync_ratio=5 r2=$L::k5_gsync_ratio=2 --rounds 8 2>&1 | grep -v amdgpu | tail -2 | sed "s/^/wan22 bf16 r2: /"
RSA_PERF_WORKLOAD=cogvideox_768p_81f RSA_PERF_REGIME=r2 python tools/ab_libs.py free=$L::k5_gsync=1 aligned=$L::k5_gsync=3 --rounds 8 2>&1 | grep -v amdgpu | tail -2 | sed "s/^/cogvideox bf16 (32-row kernel) r2: /"
