#!/bin/bash
# round 4 mid-way check: the new processor tests, the exchange tests, the default bench line, two ranks on one device
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_processors_r4.py tests/test_gpu_shard_invariance.py tests/test_gpu_select_paths.py tests/test_gpu_fullsize.py -q -m gpu -x ) > gpurun_out/r4n_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r4n_tests.txt
( timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r4n_bench.json 2> gpurun_out/r4n_bench.err
( RSA_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-extras --no-cpu-baseline | grep "^{" ) > gpurun_out/r4n_bench_2ranks.json 2> gpurun_out/r4n_bench_2ranks.err
grep -E "passed|failed|rc=" gpurun_out/r4n_tests.txt | tail -3; tail -c 1500 gpurun_out/r4n_bench.json; echo; tail -3 gpurun_out/r4n_bench.err; tail -c 900 gpurun_out/r4n_bench_2ranks.json; tail -3 gpurun_out/r4n_bench_2ranks.err
