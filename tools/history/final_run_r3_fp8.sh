#!/bin/bash
# the e4m3 part of the round-3 measurement set (re-run after the hand-placed fp8 block): bench lines, kernel stats, PMC, traffic
set -x
R=$PWD
bash tools/pmc_traffic.sh r3z_pmc_r2_fp8 r2 fp8 > gpurun_out/r3z_pmc_r2_fp8.txt 2>&1
cp gpurun_out/r3z_pmc_r2_fp8/traffic.json gpurun_out/r03_k5_traffic_r2_fp8.json; cp gpurun_out/r3z_pmc_r2_fp8/traffic.json profiles/r03_k5_traffic_r2_fp8.json
rm -rf gpurun_out/r3z_pmc_r2_fp8
python bench.py --steps 20 --warmup 5 --qkv-fp8 --no-cpu-baseline > gpurun_out/r3z_bench_fp8.json 2> gpurun_out/r3z_bench_fp8.err
python bench.py --steps 20 --warmup 3 --workload wan22_ti2v_720p_121f --qkv-fp8 --no-cpu-baseline --no-extras > gpurun_out/r3z_bench_wan22_fp8.json 2>> gpurun_out/r3z_bench_fp8.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3z_prof_fp8 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --qkv-fp8 > $R/gpurun_out/r3z_prof_fp8.log 2>&1
cd $R
python3 tools/summarize_prof.py $(find gpurun_out/r3z_prof_fp8 -name "*kernel_stats.csv" | head -1) > gpurun_out/r3z_kernel_stats_fp8.md
RSA_PERF_FP8=1 bash tools/pmc_passes.sh r3z_pmc_fp8 > gpurun_out/r3z_pmc_fp8.txt 2>&1
cp gpurun_out/r3z_pmc_fp8/summary.txt gpurun_out/r3z_pmc_summary_fp8.txt; rm -rf gpurun_out/r3z_pmc_fp8
find gpurun_out/r3z_prof_fp8 -name "*kernel_trace.csv" -delete
tail -c 300 gpurun_out/r3z_bench_fp8.json
