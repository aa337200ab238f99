#!/bin/bash
# round-2 GPU batch 1: GPU tests, per-regime K5 timings + list overlap, the new bench line, traffic of locality / r2
set -x
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r2a_tests.txt
python tools/perf_k5.py regimes > gpurun_out/r2a_regimes.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
bash tools/pmc_traffic.sh r2a_pmc_loc locality > gpurun_out/r2a_pmc_loc.txt 2>&1
bash tools/pmc_traffic.sh r2a_pmc_r2 r2 > gpurun_out/r2a_pmc_r2.txt 2>&1
tail -3 gpurun_out/r2a_tests.txt; cat gpurun_out/r2a_regimes.txt | tail -5; tail -c 1500 gpurun_out/r2a_bench.json
