#!/bin/bash
# round 4: whole GPU suite with the 64-row K5 selected + regimes / rank shapes with both kernels
export RSA_TUNING=1
mkdir -p gpurun_out
( RSA_K5_W64=1 timeout 2400 python -m pytest tests -q -m gpu -x ) > gpurun_out/r4h_tests_w64.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r4h_tests_w64.txt
for W in 0 1; do
  for H in 24 3; do
    ( RSA_K5_W64=$W RSA_PERF_H=$H timeout 600 python tools/perf_k5.py regimes ) > gpurun_out/r4h_regimes_w${W}_h${H}.txt 2>&1
  done
done
tail -5 gpurun_out/r4h_tests_w64.txt | cut -c1-300; grep -h "regime" gpurun_out/r4h_regimes_w*.txt | cut -c1-140
