#!/bin/bash
# round 6: the fp8 / pv lines of the other workloads under the new form of the output check
for WL in flux_4096 wan21_720p_81f cogvideox_768p_81f; do
  for M in 1 pv; do
    python bench.py --steps 10 --warmup 3 --workload $WL --qkv-fp8 $M --no-cpu-baseline --no-extras > gpurun_out/r6u_bench_${WL}_$M.json 2> gpurun_out/r6u_bench_${WL}_$M.err
    echo "$WL $M exit=$?" >> gpurun_out/r6u_exits.txt
  done
done
