#!/bin/bash
set -x
R=$PWD
python -m pytest tests/test_gpu_fp8.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version" | tail -8 > gpurun_out/r3_chk_tests.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_chk_prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --qkv-fp8 > $R/gpurun_out/r3_chk_prof.log 2>&1
cd $R
python3 tools/summarize_prof.py $(find gpurun_out/r3_chk_prof -name "*kernel_stats.csv" | head -1) > gpurun_out/r3_chk_kernel_stats.md
find gpurun_out/r3_chk_prof -name "*kernel_trace.csv" -delete
tail -5 gpurun_out/r3_chk_tests.txt
