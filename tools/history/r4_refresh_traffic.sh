#!/bin/bash
# re-collect the committed fall-back traffic jsons (and the bench kernel stats) with the final kernel sources
export RSA_TUNING=1
for RG in r2 r1 locality; do
  bash tools/pmc_traffic.sh r4y_pmc_$RG $RG > gpurun_out/r4y_pmc_$RG.txt 2>&1
  cp gpurun_out/r4y_pmc_$RG/traffic.json gpurun_out/r04_k5_traffic_$RG.json; rm -rf gpurun_out/r4y_pmc_$RG
done
bash tools/pmc_traffic.sh r4y_pmc_r2_fp8 r2 fp8 > gpurun_out/r4y_pmc_r2_fp8.txt 2>&1
cp gpurun_out/r4y_pmc_r2_fp8/traffic.json gpurun_out/r04_k5_traffic_r2_fp8.json; rm -rf gpurun_out/r4y_pmc_r2_fp8
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4y_prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/r4y_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4y_prof_fp8 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --qkv-fp8 > $R/gpurun_out/r4y_prof_fp8.log 2>&1
RSA_PERF_H=3 RSA_PERF_REGIMES=r2 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4y_prof_h3 -- python3 $R/tools/perf_k5.py regimes > $R/gpurun_out/r4y_prof_h3.log 2>&1
cd $R
for T in prof prof_fp8 prof_h3; do
  F=$(find gpurun_out/r4y_$T -name "*kernel_stats.csv" | head -1)
  cp $F gpurun_out/r4y_${T}_kernel_stats.csv
  python3 tools/summarize_prof.py $F > gpurun_out/r4y_${T}_kernel_stats.md
  rm -rf gpurun_out/r4y_$T
done
grep -h sha gpurun_out/r04_k5_traffic_*.json | sort | uniq -c; sed -n 7,9p gpurun_out/r4y_prof_kernel_stats.md | cut -c1-120
