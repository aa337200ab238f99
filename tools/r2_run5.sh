#!/bin/bash
set -x
timeout 600 python -m pytest tests/test_gpu_pairs.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r2g_tests.txt
cat gpurun_out/r2g_tests.txt
RSA_PERF_REGIMES=locality,r2 timeout 600 python tools/perf_k5.py pair > gpurun_out/r2g_pair.txt 2>&1
cat gpurun_out/r2g_pair.txt
