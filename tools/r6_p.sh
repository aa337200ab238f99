#!/bin/bash
# round 6: in-kernel stamps of the head-dim-64 64-row K5 (diag build, CogVideoX R2)
mkdir -p gpurun_out
make -s -C rectified_spaattn_amd/csrc diag > /dev/null 2>&1
RSA_PERF_WORKLOAD=cogvideox_768p_81f python tools/diag_k5w.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6p_diag_d64.txt; cat gpurun_out/r6p_diag_d64.txt | cut -c1-800
