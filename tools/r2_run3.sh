#!/bin/bash
set -x
RSA_PERF_REGIMES=r2 RSA_PERF_OPTS=0,2,1,194,514,642,66 timeout 600 python tools/perf_k5.py pp > gpurun_out/r2d_pp.txt 2>&1
cat gpurun_out/r2d_pp.txt
