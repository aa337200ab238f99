#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_layouts.py tests/test_gpu_select_paths.py tests/test_gpu_fullsize.py tests/test_gpu_api.py -x -q 2>&1 | tail -8 > gpurun_out/r2r_tests.txt
cat gpurun_out/r2r_tests.txt
bash tools/prof_quick.sh r2r_prof --no-extras 2>&1 | head -30
rm -rf gpurun_out/r2r_prof
