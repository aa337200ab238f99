#!/bin/bash
set -x
RSA_PERF_OPTS=1,4,1,4 timeout 600 python tools/perf_k5.py variants > gpurun_out/r2h_var.txt 2>&1
cat gpurun_out/r2h_var.txt
