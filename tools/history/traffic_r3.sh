#!/bin/bash
# memory-side traffic of K5 (PMC passes) in the three regimes + the e4m3 kernel, stamped with the current kernel-source sha;
# copies into profiles/ on the box so that the bench line that follows picks them up
set -x
for RG in r2 r1 locality; do
  bash tools/pmc_traffic.sh r3t_pmc_$RG $RG > gpurun_out/r3t_pmc_$RG.txt 2>&1
  cp gpurun_out/r3t_pmc_$RG/traffic.json gpurun_out/r03_k5_traffic_$RG.json
  cp gpurun_out/r3t_pmc_$RG/traffic.json profiles/r03_k5_traffic_$RG.json
  rm -rf gpurun_out/r3t_pmc_$RG
done
bash tools/pmc_traffic.sh r3t_pmc_r2_fp8 r2 fp8 > gpurun_out/r3t_pmc_r2_fp8.txt 2>&1
cp gpurun_out/r3t_pmc_r2_fp8/traffic.json gpurun_out/r03_k5_traffic_r2_fp8.json; cp gpurun_out/r3t_pmc_r2_fp8/traffic.json profiles/r03_k5_traffic_r2_fp8.json
rm -rf gpurun_out/r3t_pmc_r2_fp8
python bench.py --steps 20 --warmup 5 --via-api > gpurun_out/r3t_bench.json 2> gpurun_out/r3t_bench.err
python bench.py --steps 20 --warmup 5 --qkv-fp8 --no-cpu-baseline > gpurun_out/r3t_bench_fp8.json 2>> gpurun_out/r3t_bench.err
tail -c 300 gpurun_out/r3t_bench.json
