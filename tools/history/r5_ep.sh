#!/bin/bash
export RSA_TUNING=1
python -m pytest tests/test_gpu_gsync.py tests/test_gpu_parity.py tests/test_gpu_api.py -x -q 2>&1 | tail -4 > gpurun_out/r5d_tests.txt
cat gpurun_out/r5d_tests.txt
L=rectified_spaattn_amd/librsa_hip.so
python tools/ab_libs.py rows=$L::k5_ep_lds=0 lds=$L::k5_ep_lds=1 --rounds 10 2>&1 | grep -v amdgpu > gpurun_out/r5d_ab_ep.txt
tail -4 gpurun_out/r5d_ab_ep.txt
RSA_PERF_H=3 python tools/ab_libs.py rows=$L::k5_ep_lds=0 lds=$L::k5_ep_lds=1 --rounds 10 2>&1 | grep -v amdgpu > gpurun_out/r5d_ab_ep_h3.txt
tail -3 gpurun_out/r5d_ab_ep_h3.txt
