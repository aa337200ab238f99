#!/bin/bash
# round 6, fourth lease: the 256-row dense form + the static reference: tests, dense timings on / off, the whole suite, the bench line
mkdir -p gpurun_out
export RSA_TUNING=1
( timeout 900 python -m pytest tests/test_gpu_rows256.py tests/test_gpu_static_reference.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -30 ) > gpurun_out/r6d_newtests.txt 2>&1
tail -12 gpurun_out/r6d_newtests.txt
L=rectified_spaattn_amd/librsa_hip.so
( timeout 900 python tools/ab_libs.py r256=$L::k5_rows256=1,k5_static=1 r128=$L::k5_rows256=0,k5_static=1 r128online=$L::k5_rows256=0,k5_static=0 --rounds 6 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6d_rows256_ab.txt
tail -4 gpurun_out/r6d_rows256_ab.txt | cut -c1-220
( timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -12 ) > gpurun_out/r6d_suite.txt 2>&1
tail -5 gpurun_out/r6d_suite.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6d_bench.json 2> gpurun_out/r6d_bench.err; tail -c 300 gpurun_out/r6d_bench.json; tail -3 gpurun_out/r6d_bench.err
