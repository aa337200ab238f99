#!/bin/bash
# re-collects the committed K5 traffic jsons (profiles/r05_k5_traffic_*.json, stamped with the kernel-source sha bench.py checks)
# after a change to the K5 sources, plus the bench lines of the workload that change touched; one lease
set -x
export RSA_TUNING=1
for RG in r2 r1 locality script; do
  bash tools/pmc_traffic.sh r5s_pmc_$RG $RG > gpurun_out/r5s_pmc_$RG.txt 2>&1
  cp gpurun_out/r5s_pmc_$RG/traffic.json gpurun_out/r05_k5_traffic_$RG.json; rm -rf gpurun_out/r5s_pmc_$RG
done
for F in fp8 pv; do
  bash tools/pmc_traffic.sh r5s_pmc_r2_$F r2 $F > gpurun_out/r5s_pmc_r2_$F.txt 2>&1
  cp gpurun_out/r5s_pmc_r2_$F/traffic.json gpurun_out/r05_k5_traffic_r2_$F.json; rm -rf gpurun_out/r5s_pmc_r2_$F
done
for M in "" "--qkv-fp8" "--qkv-fp8 pv"; do
  T=$(echo "$M" | sed 's/--qkv-fp8 pv/_pv/; s/--qkv-fp8/_fp8/')
  python bench.py --steps 20 --warmup 3 --workload cogvideox_768p_81f $M --no-cpu-baseline --no-extras > gpurun_out/r5s_bench_cogvideox$T.json 2>> gpurun_out/r5s_bench.err
done
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -3 > gpurun_out/r5s_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5s_smoke.txt 2>&1
cat gpurun_out/r5s_tests.txt; tail -1 gpurun_out/r5s_smoke.txt
