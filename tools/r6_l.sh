#!/bin/bash
# round 6: flakiness check on the final tree: the GPU suite twice more, the randomized sweeps with other seeds
mkdir -p gpurun_out
export RSA_TUNING=1
for R in 1 2; do ( timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -3 ) > gpurun_out/r6l_suite_$R.txt 2>&1; cat gpurun_out/r6l_suite_$R.txt; done
( timeout 1500 python tests/diag/sweep_dense.py 71 200 2>&1 | tail -3 ) > gpurun_out/r6l_sweep_dense.txt; cat gpurun_out/r6l_sweep_dense.txt
( timeout 1800 python tests/diag/sweep_random_layouts.py 72 160 2>&1 | tail -3 ) > gpurun_out/r6l_sweep_layouts.txt; cat gpurun_out/r6l_sweep_layouts.txt
