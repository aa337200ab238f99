#!/bin/bash
# Round-4 measurement set: GPU tests, smoke, memory-side traffic of K5 per regime (PMC), the default bench line (all sub-records),
# per-workload lines, rocprofv3 kernel stats (raw CSV kept), PMC summaries (K5 64-row, K5 e4m3, select pass, 3-head shape),
# interleaved A/B per regime (free-running walks / aligned starts / the 32-row kernel), L2 hit / miss counters of both settings,
# what each part of the loop costs (the A/B forms), the probes, in-kernel stamps, two ranks on one device.  Everything lands in
# gpurun_out/r4z_*.
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
FREE=k5_w64=1,k5_gsync=0,k5_text_last=0; ALN=k5_w64=1,k5_gsync=1,k5_text_last=1; R32=k5_w64=0,k5_gsync=0,k5_text_last=0
for RG in r2 r1 locality; do
  RSA_PERF_REGIME=$RG python tools/ab_libs.py free=$L::$FREE aligned=$L::$ALN row32=$L::$R32 --rounds 10 > gpurun_out/r4z_ab_$RG.txt 2>&1
done
python tools/ab_libs.py free=$L::k5_gsync=0 aligned=$L::k5_gsync=3 --rounds 4 --fp8 > gpurun_out/r4z_ab_fp8.txt 2>&1
# the free-running walks' counters beside the product's (pmc_summary above)
RSA_K5_GSYNC=0 bash tools/pmc_passes.sh r4z_pmc_free > gpurun_out/r4z_pmc_free.txt 2>&1
cp gpurun_out/r4z_pmc_free/summary.txt gpurun_out/r4z_pmc_summary_free.txt; rm -rf gpurun_out/r4z_pmc_free
( cd /tmp
  for G in 0 1; do
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/r4z_l2_$G -- python3 $R/tools/ab_libs.py x=$R/$L::k5_w64=1,k5_gsync=$G --pmc > /dev/null 2>&1
  done )
python3 tools/l2_hits.py gpurun_out/r4z_l2_0 gpurun_out/r4z_l2_1 > gpurun_out/r4z_l2.txt 2>&1; rm -rf gpurun_out/r4z_l2_0 gpurun_out/r4z_l2_1
# what each part of the loop costs: forms of librsa_hip_ab.so with work removed (0 = product, 1 no staging, 3 no exponentials,
# 4 no vector work, 5 MFMAs + LDS reads, 8 no staging / boundary), free-running and aligned
LA=rectified_spaattn_amd/librsa_hip_ab.so
for G in 0 1; do
  S=""; for n in 0 1 3 4 5 8; do S="$S x$n=$LA::k5w_form=$n,k5_gsync=$G"; done
  python tools/ab_libs.py $S --rounds 4 > gpurun_out/r4z_forms_gsync$G.txt 2>&1
done
tools/probes/dma_issue_probe2 > gpurun_out/r4z_dma_probe2.txt 2>&1
tools/probes/k5w_block_probe > gpurun_out/r4z_block_probe.txt 2>&1
python tools/diag_k5w.py > gpurun_out/r4z_diag.txt 2>&1
tools/probes/dma_issue_probe > gpurun_out/r4z_dma_probe.txt 2>&1
RSA_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 2 --no-extras --no-cpu-baseline | grep "^{" > gpurun_out/r4z_bench_2ranks_one_device.json 2>> gpurun_out/r4z_bench.err
for CFG in "1 1" "1 0" "0 0"; do set -- $CFG; for HH in 24 12 6 3; do echo "k5_w64=$1 k5_gsync=$2 heads=$HH"; RSA_K5_W64=$1 RSA_K5_GSYNC=$2 RSA_PERF_H=$HH RSA_PERF_REGIMES=r2 python tools/perf_k5.py regimes; done; done > gpurun_out/r4z_rank_shapes.txt 2>&1
python tools/clock_probe.py > gpurun_out/r4z_clock.txt 2>&1
du -sh gpurun_out
tail -3 gpurun_out/r4z_tests.txt; cat gpurun_out/r4z_smoke.txt | tail -2; tail -c 600 gpurun_out/r4z_bench.json
