#!/bin/bash
export RSA_TUNING=1
python -m pytest tests/test_gpu_select_paths.py -x -q 2>&1 | tail -8 > gpurun_out/r5c_tests.txt
cat gpurun_out/r5c_tests.txt
