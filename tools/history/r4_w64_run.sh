#!/bin/bash
# round 4: the 64-rows-per-wave K5 on the hardware: smoke + parity subsets with the new kernel selected, the interleaved A/B
# against the product kernel (one process, one device), in-kernel stamps
export RSA_TUNING=1
mkdir -p gpurun_out
T=${1:-r4c}
( RSA_K5_W64=1 timeout 300 python __graft_entry__.py smoke ) > gpurun_out/${T}_smoke.txt 2>&1
echo "smoke rc=$?" >> gpurun_out/${T}_smoke.txt
( RSA_K5_W64=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_random_layouts.py -x -q -m gpu ) > gpurun_out/${T}_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/${T}_tests.txt
L=rectified_spaattn_amd/librsa_hip.so
( timeout 600 python tools/ab_libs.py cur=$L::k5_w64=0 w64=$L::k5_w64=1 --rounds 6 ) > gpurun_out/${T}_ab.txt 2>&1
echo "ab rc=$?" >> gpurun_out/${T}_ab.txt
( timeout 300 python tools/diag_k5w.py ) > gpurun_out/${T}_diag.txt 2>&1
tail -3 gpurun_out/${T}_smoke.txt; tail -12 gpurun_out/${T}_tests.txt | cut -c1-300; tail -4 gpurun_out/${T}_ab.txt; tail -2 gpurun_out/${T}_diag.txt
