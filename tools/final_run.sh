set -x
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/final_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.txt 2>&1
python bench.py --steps 10 --warmup 3 > gpurun_out/bench_final_bf16.json 2> gpurun_out/bench_final.err
python bench.py --steps 10 --warmup 3 --qkv-fp8 --no-cpu-baseline > gpurun_out/bench_final_fp8.json 2>> gpurun_out/bench_final.err
python bench.py --steps 10 --warmup 3 --neighbors gilbert --p-remain 0.05 --no-cpu-baseline > gpurun_out/bench_final_r1.json 2>> gpurun_out/bench_final.err
python bench.py --steps 20 --warmup 3 --workload wan22_ti2v_720p_121f --qkv-fp8 --no-cpu-baseline > gpurun_out/bench_final_wan22_fp8.json 2>> gpurun_out/bench_final.err
python bench.py --steps 20 --warmup 3 --workload wan22_ti2v_720p_121f --no-cpu-baseline > gpurun_out/bench_final_wan22_bf16.json 2>> gpurun_out/bench_final.err
python bench.py --steps 10 --warmup 3 --workload flux_4096 --no-cpu-baseline > gpurun_out/bench_final_flux.json 2>> gpurun_out/bench_final.err
python bench.py --steps 10 --warmup 3 --workload wan21_720p_81f --no-cpu-baseline > gpurun_out/bench_final_wan21.json 2>> gpurun_out/bench_final.err
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof6 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof6.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof7 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --qkv-fp8 > $R/gpurun_out/prof7.log 2>&1
cd $R; find gpurun_out/prof6 gpurun_out/prof7 -name "*kernel_stats*"
