#!/bin/bash
set -x
python -m pytest tests/test_gpu_fp8.py tests/test_gpu_fullsize.py tests/test_gpu_api.py -x -q -m gpu -s 2>&1 | grep -v "^RCCL\|^HIP version" | grep "code-map\|passed\|failed\|Error\|assert" > gpurun_out/r3_chk_tests.txt
tail -5 gpurun_out/r3_chk_tests.txt
