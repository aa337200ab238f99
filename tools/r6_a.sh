#!/bin/bash
# round 6, first lease: the new tests, then the whole suite + smoke + bench line, then the timing-only forms 18-21
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_headline_launch.py tests/test_gpu_masked.py tests/test_gpu_shard_invariance.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -25 ) > gpurun_out/r6a_newtests.txt 2>&1
cat gpurun_out/r6a_newtests.txt | tail -15
bash tools/gpu_suite.sh 2>/dev/null
cp gpurun_out/suite_tests.txt gpurun_out/r6a_suite_tests.txt; cp gpurun_out/suite_bench.json gpurun_out/r6a_bench.json
FORMS="0 18 19 20 21 15" ROUNDS=5 bash tools/r6_forms.sh r6f2
