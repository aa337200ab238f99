#!/bin/bash
set -x
timeout 900 python -m pytest tests/test_gpu_select_paths.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_random_layouts.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r2i_tests.txt
cat gpurun_out/r2i_tests.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r2i_prof -- python3 /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /root/repo/gpurun_out/r2i_prof.log 2>&1
cd /root/repo; python3 tools/summarize_prof.py $(find gpurun_out/r2i_prof -name "*kernel_stats.csv" | head -1) | cut -c1-150
