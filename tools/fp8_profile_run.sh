set -x
python bench.py --qkv-fp8 --steps 10 --warmup 3 > gpurun_out/bench_fp8_hunyuan.json 2> gpurun_out/bench_fp8.err
python bench.py --qkv-fp8 --workload wan22_ti2v_720p_121f --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_fp8_wan22.json 2>> gpurun_out/bench_fp8.err
python bench.py --workload wan22_ti2v_720p_121f --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_bf16_wan22.json 2>> gpurun_out/bench_fp8.err
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof5 -- python3 $R/bench.py --qkv-fp8 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof5.log 2>&1
cd $R; find gpurun_out/prof5 -name "*kernel_stats*" | head
