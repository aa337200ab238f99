#!/bin/bash
# the round-end checks the driver runs, on one leased GPU: full GPU suite, smoke, the default bench line
set -x
python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -6 > gpurun_out/suite_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > gpurun_out/suite_smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/suite_bench.json 2> gpurun_out/suite_bench.err
cat gpurun_out/suite_tests.txt gpurun_out/suite_smoke.txt; tail -c 400 gpurun_out/suite_bench.json; tail -3 gpurun_out/suite_bench.err
