#!/bin/bash
export RSA_TUNING=1
python -m pytest tests/test_gpu_select_paths.py -x -q -k "k2_forms" 2>&1 | tail -5 > gpurun_out/r5b_tests.txt
cat gpurun_out/r5b_tests.txt
python tools/perf_select.py k2_split=0,1,2 2>&1 | grep -v amdgpu.ids > gpurun_out/r5b_select.txt
RSA_PERF_H=3 python tools/perf_select.py k2_split=0,1,2 2>&1 | grep -v amdgpu.ids  >> gpurun_out/r5b_select.txt
RSA_PERF_WORKLOAD=wan21_720p_81f python tools/perf_select.py k2_split=0,1,2  2>&1 | grep -v amdgpu.ids >> gpurun_out/r5b_select.txt
cat gpurun_out/r5b_select.txt
