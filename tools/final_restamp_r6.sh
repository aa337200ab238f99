#!/bin/bash
# Round 6, after the head-dim-64 kernel: what changed since tools/final_run_r6.sh ran -- the GPU suite and smoke on the final tree, the K5
# traffic jsons re-collected (they are stamped with the kernel-source sha bench.py checks), the default bench line, the CogVideoX lines
# (2-byte on the 64-row kernel now), rocprofv3 kernel stats of the default line and of the CogVideoX line.  Everything in gpurun_out/r6y_*.
set -x
export RSA_TUNING=1
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 > gpurun_out/r6y_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6y_smoke.txt 2>&1
for RG in r2 r1 locality script; do
  bash tools/pmc_traffic.sh r6y_pmc_$RG $RG > gpurun_out/r6y_pmc_$RG.txt 2>&1
  cp gpurun_out/r6y_pmc_$RG/traffic.json gpurun_out/r06_k5_traffic_$RG.json
  cp gpurun_out/r6y_pmc_$RG/traffic.json profiles/r06_k5_traffic_$RG.json
  rm -rf gpurun_out/r6y_pmc_$RG
done
for F in fp8 pv; do
  bash tools/pmc_traffic.sh r6y_pmc_r2_$F r2 $F > gpurun_out/r6y_pmc_r2_$F.txt 2>&1
  cp gpurun_out/r6y_pmc_r2_$F/traffic.json gpurun_out/r06_k5_traffic_r2_$F.json; cp gpurun_out/r6y_pmc_r2_$F/traffic.json profiles/r06_k5_traffic_r2_$F.json
  rm -rf gpurun_out/r6y_pmc_r2_$F
done
python bench.py --steps 20 --warmup 5 > gpurun_out/r6y_bench.json 2> gpurun_out/r6y_bench.err
for M in "" "--qkv-fp8" "--qkv-fp8 pv"; do
  T=$(echo "$M" | sed 's/--qkv-fp8 pv/_pv/; s/--qkv-fp8/_fp8/')
  python bench.py --steps 20 --warmup 3 --workload cogvideox_768p_81f $M --no-cpu-baseline --no-live-traffic > gpurun_out/r6y_bench_cogvideox$T.json 2>> gpurun_out/r6y_bench.err
done
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6y_prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/r6y_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6y_prof_cog -- python3 $R/bench.py --steps 20 --warmup 3 --workload cogvideox_768p_81f --no-cpu-baseline --no-extras > $R/gpurun_out/r6y_prof_cog.log 2>&1
cd $R
for T in prof prof_cog; do
  F=$(find gpurun_out/r6y_$T -name "*kernel_stats.csv" | head -1)
  cp $F gpurun_out/r6y_${T}_kernel_stats.csv
  python3 tools/summarize_prof.py $F > gpurun_out/r6y_${T}_kernel_stats.md
  find gpurun_out/r6y_$T -name "*kernel_trace.csv" -delete
done
tail -3 gpurun_out/r6y_tests.txt; tail -1 gpurun_out/r6y_smoke.txt; tail -c 400 gpurun_out/r6y_bench.json
