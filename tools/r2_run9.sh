#!/bin/bash
set -x
timeout 900 python -m pytest tests/test_gpu_shard_invariance.py tests/test_gpu_c_host.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r2k_tests.txt
cat gpurun_out/r2k_tests.txt
