#!/bin/bash
# round 4, first contact of the 64-rows-per-wave K5 with the hardware: smoke + parity subsets with the new kernel selected,
# then the interleaved A/B against the product kernel (one process, one device)
export RSA_TUNING=1
mkdir -p gpurun_out
( RSA_K5_W64=1 timeout 300 python __graft_entry__.py smoke ) > gpurun_out/r4a_smoke.txt 2>&1
echo "smoke rc=$?" >> gpurun_out/r4a_smoke.txt
( RSA_K5_W64=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_random_layouts.py -x -q -m gpu ) > gpurun_out/r4a_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r4a_tests.txt
L=rectified_spaattn_amd/librsa_hip.so
( timeout 600 python tools/ab_libs.py cur=$L::k5_w64=0 w64=$L::k5_w64=1 --rounds 6 ) > gpurun_out/r4a_ab.txt 2>&1
echo "ab rc=$?" >> gpurun_out/r4a_ab.txt
tail -5 gpurun_out/r4a_smoke.txt; tail -15 gpurun_out/r4a_tests.txt; tail -12 gpurun_out/r4a_ab.txt
