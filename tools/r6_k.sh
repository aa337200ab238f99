#!/bin/bash
# round 6: the walk with the partial last block first and the loop taking the last full block (no C++-driven tail): whole suite, then A/B against
# the previous build (librsa_hip_prev.so, built from the commit before) on R2, the script regime, Flux, Wan2.2; dense 16k in the same table
mkdir -p gpurun_out
export RSA_TUNING=1
( timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -12 ) > gpurun_out/r6k_suite.txt 2>&1
tail -5 gpurun_out/r6k_suite.txt
L=rectified_spaattn_amd/librsa_hip.so; P=rectified_spaattn_amd/librsa_hip_prev.so
( timeout 900 python tools/ab_libs.py new=$L prev=$P --rounds 8 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6k_ab_r2.txt; tail -3 gpurun_out/r6k_ab_r2.txt | cut -c1-200
( RSA_PERF_REGIME=script timeout 900 python tools/ab_libs.py new=$L prev=$P --rounds 4 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6k_ab_script.txt; tail -2 gpurun_out/r6k_ab_script.txt | cut -c1-200
for W in flux_4096 wan22_ti2v_720p_121f; do
  ( RSA_PERF_WORKLOAD=$W timeout 900 python tools/ab_libs.py new=$L prev=$P --rounds 6 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6k_ab_$W.txt; tail -2 gpurun_out/r6k_ab_$W.txt | cut -c1-200
done
