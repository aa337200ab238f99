#!/bin/bash
# round 6: 256-row dense tiles at head dim 64 (tests, dense CogVideoX-shape timing on / off), then the randomized sweeps of tests/diag on the
# final kernels (dense calls and random layouts against the oracle)
mkdir -p gpurun_out
export RSA_TUNING=1
( timeout 900 python -m pytest tests/test_gpu_rows256.py tests/test_gpu_static_reference.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -12 ) > gpurun_out/r6j_tests.txt 2>&1
tail -4 gpurun_out/r6j_tests.txt
for R in 1 0 1 0; do RSA_K5_ROWS256=$R python tools/perf_d64.py 2>&1 | grep -v amdgpu.ids | sed "s/^/rows256=$R: /"; done > gpurun_out/r6j_d64_dense.txt; cat gpurun_out/r6j_d64_dense.txt
( timeout 1200 python tests/diag/sweep_dense.py 61 150 2>&1 | tail -4 ) > gpurun_out/r6j_sweep_dense.txt; cat gpurun_out/r6j_sweep_dense.txt
( timeout 1500 python tests/diag/sweep_random_layouts.py 62 120 2>&1 | tail -4 ) > gpurun_out/r6j_sweep_layouts.txt; cat gpurun_out/r6j_sweep_layouts.txt
