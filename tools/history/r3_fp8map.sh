#!/bin/bash
# A/B of the e4m3 kernel's P forms (fp8_variant 0 = code map hand-placed, 1 = its compiled twin, 2 = exact exponential) + the fp8 tests
set -x
RSA_PERF_OPTS=0,1,2 python tools/perf_k5.py fp8variants > gpurun_out/r3_fp8map_ab.txt 2>&1
python -m pytest tests/test_gpu_fp8.py -x -q -m gpu > gpurun_out/r3_fp8map_tests.txt 2>&1
tail -5 gpurun_out/r3_fp8map_tests.txt
cat gpurun_out/r3_fp8map_ab.txt
