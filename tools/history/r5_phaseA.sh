#!/bin/bash
# round 5, first lease: the new tests, the default bench line with the script regime, aligned-start guard A/B at the script point
set -x
export RSA_TUNING=1
python -m pytest tests/test_gpu_tail_split.py tests/test_gpu_shard_invariance.py tests/test_gpu_c_host.py tests/test_gpu_gsync.py tests/test_gpu_api.py -x -q 2>&1 | tail -8 > gpurun_out/r5a_tests.txt
python -m pytest tests/test_gpu_fullsize.py -x -q -k "whole_head" 2>&1 | tail -8 >> gpurun_out/r5a_tests.txt
cat gpurun_out/r5a_tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err
tail -c 1500 gpurun_out/r5a_bench.json; tail -3 gpurun_out/r5a_bench.err
L=rectified_spaattn_amd/librsa_hip.so
RSA_PERF_REGIME=script python tools/ab_libs.py r5=$L::k5_gsync_ratio=5 r4=$L::k5_gsync_ratio=4 r3=$L::k5_gsync_ratio=3 r2=$L::k5_gsync_ratio=2 --rounds 6 > gpurun_out/r5a_ab_script.txt 2>&1
tail -8 gpurun_out/r5a_ab_script.txt
RSA_PERF_WORKLOAD=wan21_720p_81f RSA_PERF_REGIME=script python tools/ab_libs.py r5=$L::k5_gsync_ratio=5 r3=$L::k5_gsync_ratio=3 r2=$L::k5_gsync_ratio=2 --rounds 4 > gpurun_out/r5a_ab_script_wan21.txt 2>&1
tail -6 gpurun_out/r5a_ab_script_wan21.txt
