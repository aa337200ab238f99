#!/bin/bash
# round 6: the smooth-K sample kernel with its blocks pooled in parallel: fp8 tests (byte-exact mu), its time under rocprofv3
python -m pytest tests/test_gpu_fp8.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3 > gpurun_out/r6z2_tests.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6z2_prof -- python3 $R/bench.py --steps 40 --warmup 5 --workload wan22_ti2v_720p_121f --qkv-fp8 1 --no-cpu-baseline --no-extras --no-live-traffic > $R/gpurun_out/r6z2_prof.log 2>&1
cd $R
F=$(find gpurun_out/r6z2_prof -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_prof.py $F > gpurun_out/r6z2_config5_e4m3_kernel_stats.md; cp $F gpurun_out/r6z2_config5_e4m3_kernel_stats.csv
rm -rf gpurun_out/r6z2_prof
