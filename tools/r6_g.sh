#!/bin/bash
# round 6: K2 with LDS-DMA double-buffered tiles (16- and 32-k chunks) against the register-staged form: tests, per-kernel times
mkdir -p gpurun_out
export RSA_TUNING=1
( timeout 900 python -m pytest tests/test_gpu_select_paths.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -30 ) > gpurun_out/r6g_newtests.txt 2>&1
tail -4 gpurun_out/r6g_newtests.txt
( python tools/perf_select.py k2_dma=16,32,0; RSA_PERF_H=3 python tools/perf_select.py k2_dma=16,32,0; RSA_PERF_WORKLOAD=wan22_ti2v_720p_121f python tools/perf_select.py k2_dma=16,32,0; RSA_PERF_WORKLOAD=cogvideox_768p_81f python tools/perf_select.py k2_dma=16,32,0; RSA_PERF_WORKLOAD=wan21_720p_81f python tools/perf_select.py k2_dma=16,32,0 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6g_select2.txt
cat gpurun_out/r6g_select2.txt
