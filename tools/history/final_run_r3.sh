#!/bin/bash
# Round-3 measurement set: GPU tests, smoke, the bench line (all sub-records), per-workload lines, rocprofv3 kernel stats,
# PMC summary of K5 and memory-side traffic of K5 in the three regimes.  Everything lands in gpurun_out/r3z_*.
set -x
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 > gpurun_out/r3z_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3z_smoke.txt 2>&1
for RG in r2 r1 locality; do
  bash tools/pmc_traffic.sh r3z_pmc_$RG $RG > gpurun_out/r3z_pmc_$RG.txt 2>&1
  cp gpurun_out/r3z_pmc_$RG/traffic.json gpurun_out/r03_k5_traffic_$RG.json
  cp gpurun_out/r3z_pmc_$RG/traffic.json profiles/r03_k5_traffic_$RG.json   # the bench lines below read these (box-local copy)
  rm -rf gpurun_out/r3z_pmc_$RG
done
bash tools/pmc_traffic.sh r3z_pmc_r2_fp8 r2 fp8 > gpurun_out/r3z_pmc_r2_fp8.txt 2>&1
cp gpurun_out/r3z_pmc_r2_fp8/traffic.json gpurun_out/r03_k5_traffic_r2_fp8.json; cp gpurun_out/r3z_pmc_r2_fp8/traffic.json profiles/r03_k5_traffic_r2_fp8.json
rm -rf gpurun_out/r3z_pmc_r2_fp8
python bench.py --steps 20 --warmup 5 --via-api > gpurun_out/r3z_bench.json 2> gpurun_out/r3z_bench.err
python bench.py --steps 20 --warmup 5 --qkv-fp8 --no-cpu-baseline > gpurun_out/r3z_bench_fp8.json 2>> gpurun_out/r3z_bench.err
python bench.py --steps 20 --warmup 3 --workload flux_4096 --no-cpu-baseline --no-extras > gpurun_out/r3z_bench_flux.json 2>> gpurun_out/r3z_bench.err
python bench.py --steps 20 --warmup 3 --workload wan21_720p_81f --no-cpu-baseline --no-extras > gpurun_out/r3z_bench_wan21.json 2>> gpurun_out/r3z_bench.err
python bench.py --steps 20 --warmup 3 --workload wan22_ti2v_720p_121f --no-cpu-baseline --no-extras > gpurun_out/r3z_bench_wan22.json 2>> gpurun_out/r3z_bench.err
python bench.py --steps 20 --warmup 3 --workload wan22_ti2v_720p_121f --qkv-fp8 --no-cpu-baseline --no-extras > gpurun_out/r3z_bench_wan22_fp8.json 2>> gpurun_out/r3z_bench.err
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3z_prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/r3z_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3z_prof_fp8 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --qkv-fp8 > $R/gpurun_out/r3z_prof_fp8.log 2>&1
cd $R
python3 tools/summarize_prof.py $(find gpurun_out/r3z_prof -name "*kernel_stats.csv" | head -1) > gpurun_out/r3z_kernel_stats.md
python3 tools/summarize_prof.py $(find gpurun_out/r3z_prof_fp8 -name "*kernel_stats.csv" | head -1) > gpurun_out/r3z_kernel_stats_fp8.md
bash tools/pmc_passes.sh r3z_pmc_all > gpurun_out/r3z_pmc_all.txt 2>&1
cp gpurun_out/r3z_pmc_all/summary.txt gpurun_out/r3z_pmc_summary.txt; rm -rf gpurun_out/r3z_pmc_all
RSA_PERF_FP8=1 bash tools/pmc_passes.sh r3z_pmc_fp8 > gpurun_out/r3z_pmc_fp8.txt 2>&1
cp gpurun_out/r3z_pmc_fp8/summary.txt gpurun_out/r3z_pmc_summary_fp8.txt; rm -rf gpurun_out/r3z_pmc_fp8
bash tools/pmc_select.sh r3z_pmcsel > /dev/null 2>&1; cp gpurun_out/r3z_pmcsel/summary.txt gpurun_out/r3z_pmc_select.txt; rm -rf gpurun_out/r3z_pmcsel
RSA_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 2 --no-extras --no-cpu-baseline --gather-transports p2p | grep "^{" > gpurun_out/r3z_bench_2ranks_one_device.json 2>> gpurun_out/r3z_bench.err
for HH in 12 6 3; do RSA_PERF_H=$HH RSA_PERF_REGIMES=r2 python tools/perf_k5.py regimes; done > gpurun_out/r3z_rank_shapes.txt 2>&1
for d in r3z_prof r3z_prof_fp8; do find gpurun_out/$d -name "*kernel_trace.csv" -delete; done
du -sh gpurun_out
tail -3 gpurun_out/r3z_tests.txt; cat gpurun_out/r3z_smoke.txt | tail -2; tail -c 600 gpurun_out/r3z_bench.json
