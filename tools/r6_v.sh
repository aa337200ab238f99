#!/bin/bash
# round 6: per-kernel times of BASELINE config 5 (Wan2.2-TI2V 720p 121f) in its three operand forms
R=$PWD; cd /tmp; export TMPDIR=/tmp
for M in 0 pv 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6v_prof_$M -- python3 $R/bench.py --steps 40 --warmup 5 --workload wan22_ti2v_720p_121f --qkv-fp8 $M --no-cpu-baseline --no-extras --no-live-traffic > $R/gpurun_out/r6v_prof_$M.log 2>&1
done
cd $R
for M in 0 pv 1; do
  F=$(find gpurun_out/r6v_prof_$M -name "*kernel_stats.csv" | head -1)
  cp $F gpurun_out/r6v_config5_${M}_kernel_stats.csv
  python3 tools/summarize_prof.py $F > gpurun_out/r6v_config5_${M}_kernel_stats.md
  rm -rf gpurun_out/r6v_prof_$M
done
