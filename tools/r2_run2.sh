#!/bin/bash
# round-2 GPU batch 2: ping-pong K5 -- small-case correctness first (bounded by timeout), then A/B timing
set -x
RSA_K5_PP=2 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r2b_tests_pp.txt
cat gpurun_out/r2b_tests_pp.txt
if grep -q "passed" gpurun_out/r2b_tests_pp.txt && ! grep -q "failed" gpurun_out/r2b_tests_pp.txt; then
  RSA_K5_PP=2 timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r2b_tests_pp_all.txt
  cat gpurun_out/r2b_tests_pp_all.txt
  timeout 600 python tools/perf_k5.py pp > gpurun_out/r2b_pp.txt 2>&1
  cat gpurun_out/r2b_pp.txt
fi
timeout 300 python -m pytest tests/test_gpu_processors_r2.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r2b_tests_new.txt
cat gpurun_out/r2b_tests_new.txt
