#!/bin/bash
export RSA_TUNING=1
python -m pytest tests/test_gpu_masked.py tests/test_gpu_api.py -x -q 2>&1 | tail -6
