#!/bin/bash
# Round-6 measurement set on ONE lease: GPU tests, smoke, memory-side traffic of K5 per regime (PMC; incl. the script regime), the
# default bench line (all sub-records), per-workload lines (2-byte and e4m3), rocprofv3 kernel stats (raw CSV kept) incl. the 3-head
# shape and the script regime, PMC summaries (K5 64-row, K5 e4m3, select pass), interleaved A/B of the aligned-start guard, rank
# shapes, the select pass per kernel (tools/perf_select.py), two ranks on one device, clocks.  Everything lands in gpurun_out/r6z_*.
set -x
export RSA_TUNING=1
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 > gpurun_out/r6z_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6z_smoke.txt 2>&1
for RG in r2 r1 locality script; do
  bash tools/pmc_traffic.sh r6z_pmc_$RG $RG > gpurun_out/r6z_pmc_$RG.txt 2>&1
  cp gpurun_out/r6z_pmc_$RG/traffic.json gpurun_out/r06_k5_traffic_$RG.json
  cp gpurun_out/r6z_pmc_$RG/traffic.json profiles/r06_k5_traffic_$RG.json   # the bench lines below read these (box-local copy)
  rm -rf gpurun_out/r6z_pmc_$RG
done
bash tools/pmc_traffic.sh r6z_pmc_r2_fp8 r2 fp8 > gpurun_out/r6z_pmc_r2_fp8.txt 2>&1
cp gpurun_out/r6z_pmc_r2_fp8/traffic.json gpurun_out/r06_k5_traffic_r2_fp8.json; cp gpurun_out/r6z_pmc_r2_fp8/traffic.json profiles/r06_k5_traffic_r2_fp8.json
rm -rf gpurun_out/r6z_pmc_r2_fp8
bash tools/pmc_traffic.sh r6z_pmc_r2_pv r2 pv > gpurun_out/r6z_pmc_r2_pv.txt 2>&1
cp gpurun_out/r6z_pmc_r2_pv/traffic.json gpurun_out/r06_k5_traffic_r2_pv.json; cp gpurun_out/r6z_pmc_r2_pv/traffic.json profiles/r06_k5_traffic_r2_pv.json
rm -rf gpurun_out/r6z_pmc_r2_pv
python bench.py --steps 20 --warmup 5 > gpurun_out/r6z_bench.json 2> gpurun_out/r6z_bench.err
python bench.py --steps 20 --warmup 5 --qkv-fp8 --no-cpu-baseline > gpurun_out/r6z_bench_fp8.json 2>> gpurun_out/r6z_bench.err
python bench.py --steps 20 --warmup 5 --qkv-fp8 pv --no-cpu-baseline > gpurun_out/r6z_bench_pv.json 2>> gpurun_out/r6z_bench.err
for WL in flux_4096 wan21_720p_81f wan22_ti2v_720p_121f cogvideox_768p_81f; do
  python bench.py --steps 20 --warmup 3 --workload $WL --no-cpu-baseline --no-live-traffic > gpurun_out/r6z_bench_$WL.json 2>> gpurun_out/r6z_bench.err
  python bench.py --steps 20 --warmup 3 --workload $WL --qkv-fp8 --no-cpu-baseline --no-extras > gpurun_out/r6z_bench_${WL}_fp8.json 2>> gpurun_out/r6z_bench.err
  python bench.py --steps 20 --warmup 3 --workload $WL --qkv-fp8 pv --no-cpu-baseline --no-extras > gpurun_out/r6z_bench_${WL}_pv.json 2>> gpurun_out/r6z_bench.err
done
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6z_prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/r6z_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6z_prof_script -- python3 $R/bench.py --steps 10 --warmup 3 --regime script --no-cpu-baseline --no-extras > $R/gpurun_out/r6z_prof_script.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6z_prof_fp8 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --qkv-fp8 > $R/gpurun_out/r6z_prof_fp8.log 2>&1
RSA_PERF_H=3 RSA_PERF_REGIMES=r2 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6z_prof_h3 -- python3 $R/tools/perf_k5.py regimes > $R/gpurun_out/r6z_prof_h3.log 2>&1
cd $R
for T in prof prof_script prof_fp8 prof_h3; do
  F=$(find gpurun_out/r6z_$T -name "*kernel_stats.csv" | head -1)
  cp $F gpurun_out/r6z_${T}_kernel_stats.csv
  python3 tools/summarize_prof.py $F > gpurun_out/r6z_${T}_kernel_stats.md
  find gpurun_out/r6z_$T -name "*kernel_trace.csv" -delete
done
bash tools/pmc_passes.sh r6z_pmc_all > gpurun_out/r6z_pmc_all.txt 2>&1
cp gpurun_out/r6z_pmc_all/summary.txt gpurun_out/r6z_pmc_summary.txt; rm -rf gpurun_out/r6z_pmc_all
RSA_PERF_REGIME=script bash tools/pmc_passes.sh r6z_pmc_script > gpurun_out/r6z_pmc_script.txt 2>&1
cp gpurun_out/r6z_pmc_script/summary.txt gpurun_out/r6z_pmc_summary_script.txt; rm -rf gpurun_out/r6z_pmc_script
RSA_PERF_FP8=1 bash tools/pmc_passes.sh r6z_pmc_fp8 > gpurun_out/r6z_pmc_fp8.txt 2>&1
cp gpurun_out/r6z_pmc_fp8/summary.txt gpurun_out/r6z_pmc_summary_fp8.txt; rm -rf gpurun_out/r6z_pmc_fp8
bash tools/pmc_select.sh r6z_pmcsel > /dev/null 2>&1; cp gpurun_out/r6z_pmcsel/summary.txt gpurun_out/r6z_pmc_select.txt; rm -rf gpurun_out/r6z_pmcsel
L=rectified_spaattn_amd/librsa_hip.so
FREE=k5_w64=1,k5_gsync=0,k5_text_last=0; ALN=k5_w64=1,k5_gsync=1,k5_text_last=1; R32=k5_w64=0,k5_gsync=0,k5_text_last=0
for RG in r2 script; do
  RSA_PERF_REGIME=$RG python tools/ab_libs.py free=$L::$FREE aligned=$L::$ALN row32=$L::$R32 --rounds 8 > gpurun_out/r6z_ab_$RG.txt 2>&1
done
ST1=k5_static=1,k5_rows256=1; ST0=k5_static=0,k5_rows256=1; R128=k5_static=1,k5_rows256=0; OLD=k5_static=0,k5_rows256=0
for RG in r2 script; do
  RSA_PERF_REGIME=$RG python tools/ab_libs.py product=$L::$ST1 online=$L::$ST0 rows128=$L::$R128 online128=$L::$OLD --rounds 8 > gpurun_out/r6z_ab_static_$RG.txt 2>&1
done
RSA_PERF_REGIMES=r2,r1,locality,script python tools/perf_k5.py regimes > gpurun_out/r6z_regimes_overlap.txt 2>&1
python tools/perf_select.py k4_split=1,0 > gpurun_out/r6z_select.txt 2>&1
RSA_PERF_H=3 python tools/perf_select.py k4_split=1,2 >> gpurun_out/r6z_select.txt 2>&1
RSA_PERF_WORKLOAD=wan22_ti2v_720p_121f python tools/perf_select.py >> gpurun_out/r6z_select.txt 2>&1
make -s -C rectified_spaattn_amd/csrc diag > /dev/null 2>&1
python tools/diag_k5w.py > gpurun_out/r6z_diag.txt 2>&1
RSA_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 2 --no-extras --no-cpu-baseline | grep "^{" > gpurun_out/r6z_bench_2ranks_one_device.json 2>> gpurun_out/r6z_bench.err
RSA_BENCH_ONE_DEVICE=1 python bench.py --gpus 8 --workload tiny8 --steps 4 --warmup 1 --no-extras --no-cpu-baseline | grep "^{" > gpurun_out/r6z_bench_8ranks_one_device.json 2>> gpurun_out/r6z_bench.err
for HH in 24 12 6 3; do echo "heads=$HH"; RSA_PERF_H=$HH RSA_PERF_REGIMES=r2 python tools/perf_k5.py regimes; done > gpurun_out/r6z_rank_shapes.txt 2>&1
python tools/perf_k5.py dense > gpurun_out/r6z_dense.txt 2>&1

python tools/clock_probe.py > gpurun_out/r6z_clock.txt 2>&1
du -sh gpurun_out
tail -3 gpurun_out/r6z_tests.txt; cat gpurun_out/r6z_smoke.txt | tail -2; tail -c 600 gpurun_out/r6z_bench.json
