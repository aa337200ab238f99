export RSA_TUNING=1
L=rectified_spaattn_amd/librsa_hip.so
timeout 300 python tools/ab_libs.py free=$L::k5_w64=1,k5_gsync=0 aligned=$L::k5_w64=1,k5_gsync=1 --rounds 4 2>&1 | tail -2 | cut -c1-140
timeout 300 python -m pytest tests/test_gpu_gsync.py tests/test_gpu_masked.py -q -m gpu 2>&1 | grep -E "passed|failed"
