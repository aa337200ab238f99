#!/bin/bash
# round 6, third lease: the optimistic static reference of the 64-row K5: its tests, the whole suite, A/B on / off in one process (R2, dense 16k)
mkdir -p gpurun_out
export RSA_TUNING=1
( timeout 900 python -m pytest tests/test_gpu_static_reference.py tests/test_gpu_select_paths.py tests/test_gpu_headline_launch.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -30 ) > gpurun_out/r6c_newtests.txt 2>&1
tail -12 gpurun_out/r6c_newtests.txt
L=rectified_spaattn_amd/librsa_hip.so
( timeout 900 python tools/ab_libs.py static=$L::k5_static=1 online=$L::k5_static=0 --rounds 8 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6c_static_ab.txt
tail -4 gpurun_out/r6c_static_ab.txt | cut -c1-220
( RSA_PERF_REGIME=script timeout 900 python tools/ab_libs.py static=$L::k5_static=1 online=$L::k5_static=0 --rounds 4 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6c_static_ab_script.txt
tail -3 gpurun_out/r6c_static_ab_script.txt | cut -c1-220
( timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -12 ) > gpurun_out/r6c_suite.txt 2>&1
tail -5 gpurun_out/r6c_suite.txt
