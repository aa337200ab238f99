#!/bin/bash
set -x
timeout 900 python -m pytest tests/test_gpu_select_paths.py tests/test_gpu_fullsize.py tests/test_gpu_api.py tests/test_gpu_c_host.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r2j_tests.txt
cat gpurun_out/r2j_tests.txt
RSA_PERF_H=3 RSA_PERF_REGIMES=r2 python tools/perf_k5.py regimes > gpurun_out/r2j_h3.txt 2>&1
RSA_TUNING=1 RSA_K5_TSPLIT=0 RSA_PERF_H=3 RSA_PERF_REGIMES=r2 python tools/perf_k5.py regimes >> gpurun_out/r2j_h3.txt 2>&1
RSA_PERF_REGIMES=r2 python tools/perf_k5.py regimes >> gpurun_out/r2j_h3.txt 2>&1
cat gpurun_out/r2j_h3.txt
