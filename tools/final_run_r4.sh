#!/bin/bash
# Round-4 measurement set: GPU tests, smoke, memory-side traffic of K5 per regime (PMC), the default bench line (all sub-records),
# per-workload lines, rocprofv3 kernel stats (raw CSV kept), PMC summaries (K5 64-row, K5 e4m3, select pass, 3-head shape),
# interleaved A/B of the two K5 kernels, in-kernel stamps, two ranks on one device.  Everything lands in gpurun_out/r4z_*.
set -x
export RSA_TUNING=1
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 > gpurun_out/r4z_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4z_smoke.txt 2>&1
for RG in r2 r1 locality; do
  bash tools/pmc_traffic.sh r4z_pmc_$RG $RG > gpurun_out/r4z_pmc_$RG.txt 2>&1
  cp gpurun_out/r4z_pmc_$RG/traffic.json gpurun_out/r04_k5_traffic_$RG.json
  cp gpurun_out/r4z_pmc_$RG/traffic.json profiles/r04_k5_traffic_$RG.json   # the bench lines below read these (box-local copy)
  rm -rf gpurun_out/r4z_pmc_$RG
done
bash tools/pmc_traffic.sh r4z_pmc_r2_fp8 r2 fp8 > gpurun_out/r4z_pmc_r2_fp8.txt 2>&1
cp gpurun_out/r4z_pmc_r2_fp8/traffic.json gpurun_out/r04_k5_traffic_r2_fp8.json; cp gpurun_out/r4z_pmc_r2_fp8/traffic.json profiles/r04_k5_traffic_r2_fp8.json
rm -rf gpurun_out/r4z_pmc_r2_fp8
python bench.py --steps 20 --warmup 5 > gpurun_out/r4z_bench.json 2> gpurun_out/r4z_bench.err
python bench.py --steps 20 --warmup 5 --qkv-fp8 --no-cpu-baseline > gpurun_out/r4z_bench_fp8.json 2>> gpurun_out/r4z_bench.err
for WL in flux_4096 wan21_720p_81f wan22_ti2v_720p_121f cogvideox_768p_81f; do
  python bench.py --steps 20 --warmup 3 --workload $WL --no-cpu-baseline --no-extras > gpurun_out/r4z_bench_$WL.json 2>> gpurun_out/r4z_bench.err
  python bench.py --steps 20 --warmup 3 --workload $WL --qkv-fp8 --no-cpu-baseline --no-extras > gpurun_out/r4z_bench_${WL}_fp8.json 2>> gpurun_out/r4z_bench.err
done
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4z_prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/r4z_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4z_prof_fp8 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --qkv-fp8 > $R/gpurun_out/r4z_prof_fp8.log 2>&1
RSA_PERF_H=3 RSA_PERF_REGIMES=r2 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4z_prof_h3 -- python3 $R/tools/perf_k5.py regimes > $R/gpurun_out/r4z_prof_h3.log 2>&1
cd $R
for T in prof prof_fp8 prof_h3; do
  F=$(find gpurun_out/r4z_$T -name "*kernel_stats.csv" | head -1)
  cp $F gpurun_out/r4z_${T}_kernel_stats.csv
  python3 tools/summarize_prof.py $F > gpurun_out/r4z_${T}_kernel_stats.md
  find gpurun_out/r4z_$T -name "*kernel_trace.csv" -delete
done
bash tools/pmc_passes.sh r4z_pmc_all > gpurun_out/r4z_pmc_all.txt 2>&1
cp gpurun_out/r4z_pmc_all/summary.txt gpurun_out/r4z_pmc_summary.txt; rm -rf gpurun_out/r4z_pmc_all
RSA_K5_W64=0 bash tools/pmc_passes.sh r4z_pmc_32row > gpurun_out/r4z_pmc_32row.txt 2>&1
cp gpurun_out/r4z_pmc_32row/summary.txt gpurun_out/r4z_pmc_summary_32row.txt; rm -rf gpurun_out/r4z_pmc_32row
RSA_PERF_FP8=1 bash tools/pmc_passes.sh r4z_pmc_fp8 > gpurun_out/r4z_pmc_fp8.txt 2>&1
cp gpurun_out/r4z_pmc_fp8/summary.txt gpurun_out/r4z_pmc_summary_fp8.txt; rm -rf gpurun_out/r4z_pmc_fp8
bash tools/pmc_select.sh r4z_pmcsel > /dev/null 2>&1; cp gpurun_out/r4z_pmcsel/summary.txt gpurun_out/r4z_pmc_select.txt; rm -rf gpurun_out/r4z_pmcsel
# where the fabric reads go (no counter separates Infinity-Cache hits from HBM: both sit behind the DRAM path)
( cd /tmp; RSA_PERF_NODENSE=1 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_GMI_32B_sum --output-format csv -d $R/gpurun_out/r4z_pmc_ea -- python3 $R/tools/perf_k5.py pmc > /dev/null 2>&1 )
python3 tools/pmc_summary.py "gpurun_out/r4z_pmc_ea/**/*counter_collection.csv" > gpurun_out/r4z_pmc_ea.txt; rm -rf gpurun_out/r4z_pmc_ea
L=rectified_spaattn_amd/librsa_hip.so
python tools/ab_libs.py row32=$L::k5_w64=0 row64=$L::k5_w64=1 --rounds 10 > gpurun_out/r4z_ab.txt 2>&1
python tools/ab_libs.py row32=$L::k5_w64=0 row64=$L::k5_w64=1 --rounds 4 --fp8 > gpurun_out/r4z_ab_fp8.txt 2>&1
python tools/diag_k5w.py > gpurun_out/r4z_diag.txt 2>&1
tools/probes/dma_issue_probe > gpurun_out/r4z_dma_probe.txt 2>&1
RSA_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 2 --no-extras --no-cpu-baseline | grep "^{" > gpurun_out/r4z_bench_2ranks_one_device.json 2>> gpurun_out/r4z_bench.err
for W in 1 0; do for HH in 24 12 6 3; do RSA_K5_W64=$W RSA_PERF_H=$HH RSA_PERF_REGIMES=r2 python tools/perf_k5.py regimes; done; done > gpurun_out/r4z_rank_shapes.txt 2>&1
python tools/clock_probe.py > gpurun_out/r4z_clock.txt 2>&1
du -sh gpurun_out
tail -3 gpurun_out/r4z_tests.txt; cat gpurun_out/r4z_smoke.txt | tail -2; tail -c 600 gpurun_out/r4z_bench.json
