#!/bin/bash
# round 6: is the lease-to-lease spread of the default line the DEVICE or the STATE earlier processes leave on it?
# the same line four times in one lease: fresh, again, after the GPU test suite (hundreds of processes / allocations / IPC), again
for T in a b; do python bench.py --no-cpu-baseline --no-extras > gpurun_out/r6x_bench_$T.json 2> gpurun_out/r6x_bench_$T.err; done
python -m pytest tests -x -q -m gpu > gpurun_out/r6x_suite.txt 2>&1
for T in c d; do python bench.py --no-cpu-baseline --no-extras > gpurun_out/r6x_bench_$T.json 2> gpurun_out/r6x_bench_$T.err; done
