#!/bin/bash
# round 6: the fp8 check of the bench lines (against the operands the kernel multiplies) + config 5 at full size
export TMPDIR=/tmp
python -m pytest tests/test_gpu_fp8.py -q -x -k "config5" -s > gpurun_out/r6s_tests.txt 2>&1
python -m pytest tests/test_gpu_headline_launch.py tests/test_gpu_static_reference.py -q -x >> gpurun_out/r6s_tests.txt 2>&1
for M in 1 pv; do
  python bench.py --workload wan22_ti2v_720p_121f --qkv-fp8 $M --no-cpu-baseline > gpurun_out/r6s_bench_wan22_$M.json 2> gpurun_out/r6s_bench_wan22_$M.err
  echo "wan22 $M exit=$?" >> gpurun_out/r6s_tests.txt
done
python bench.py --qkv-fp8 1 --no-cpu-baseline > gpurun_out/r6s_bench_fp8.json 2> gpurun_out/r6s_bench_fp8.err; echo "hunyuan fp8 exit=$?" >> gpurun_out/r6s_tests.txt
python bench.py --qkv-fp8 pv --no-cpu-baseline > gpurun_out/r6s_bench_pv.json 2> gpurun_out/r6s_bench_pv.err; echo "hunyuan pv exit=$?" >> gpurun_out/r6s_tests.txt
python bench.py > gpurun_out/r6s_bench.json 2> gpurun_out/r6s_bench.err; echo "default exit=$?" >> gpurun_out/r6s_tests.txt
