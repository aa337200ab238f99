#!/bin/bash
# round 6, second lease: new tests again, K4 split form (tests + per-kernel times, 24 and 3 heads), the valid schedule forms 22-26, the whole suite
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_headline_launch.py tests/test_gpu_select_paths.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -25 ) > gpurun_out/r6b_newtests.txt 2>&1
tail -8 gpurun_out/r6b_newtests.txt
( python tools/perf_select.py k4_split=1,0; RSA_PERF_H=3 python tools/perf_select.py k4_split=1,0 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6b_select.txt
cat gpurun_out/r6b_select.txt
FORMS="0 22 23 24 25 26" ROUNDS=6 bash tools/r6_forms.sh r6f3
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -12 ) > gpurun_out/r6b_suite.txt 2>&1
tail -5 gpurun_out/r6b_suite.txt
