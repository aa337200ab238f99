#!/bin/bash
# round 6, after the fp8 check change: the driver's round-end checks on the tree as it stands (suite, smoke, default line)
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r6t_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6t_smoke.txt 2>&1
python bench.py > gpurun_out/r6t_bench.json 2> gpurun_out/r6t_bench.err; echo "exit=$?" >> gpurun_out/r6t_tests.txt
for WL in wan22_ti2v_720p_121f; do
  python bench.py --steps 20 --warmup 3 --workload $WL --qkv-fp8 --no-cpu-baseline --no-extras > gpurun_out/r6t_bench_${WL}_fp8.json 2>> gpurun_out/r6t_bench.err
  python bench.py --steps 20 --warmup 3 --workload $WL --qkv-fp8 pv --no-cpu-baseline --no-extras > gpurun_out/r6t_bench_${WL}_pv.json 2>> gpurun_out/r6t_bench.err
done
