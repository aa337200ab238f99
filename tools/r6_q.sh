#!/bin/bash
# round 6: head dim 64 with the row sums on the matrix pipe (RSM): every head-dim-64 test, then the CogVideoX line of this build against the build
# of the commit before (librsa_hip_prev.so), alternating on one box, three rounds
mkdir -p gpurun_out
export RSA_TUNING=1
( timeout 1500 python -m pytest tests/test_gpu_select_paths.py tests/test_gpu_parity.py tests/test_gpu_gsync.py tests/test_gpu_random_layouts.py tests/test_gpu_fullsize.py tests/test_gpu_static_reference.py tests/test_gpu_api.py tests/test_gpu_rows256.py tests/test_gpu_tail_split.py -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -8 ) > gpurun_out/r6q_tests.txt 2>&1
cat gpurun_out/r6q_tests.txt
P=rectified_spaattn_amd
cp $P/librsa_hip.so /tmp/librsa_hip_product.so
for R in 1 2 3; do
for V in product prev; do
  if [ $V = product ]; then cp /tmp/librsa_hip_product.so $P/librsa_hip.so; else cp $P/librsa_hip_$V.so $P/librsa_hip.so; fi
  python bench.py --steps 20 --warmup 3 --workload cogvideox_768p_81f --no-cpu-baseline --no-extras --no-live-traffic > gpurun_out/r6q_$V.json 2>> gpurun_out/r6q.err
  python -c "import json;d=json.load(open('gpurun_out/r6q_$V.json'));print('$V round $R',d['ms_per_step'],d['roofline']['frac'],d['roofline']['k5_ms'],d['check']['ok'],d['check']['max_abs'])"
done
done 2>&1 | tee gpurun_out/r6q_variants.txt
cp /tmp/librsa_hip_product.so $P/librsa_hip.so
for R in 1 0; do RSA_K5_ROWS256=$R python tools/perf_d64.py 2>&1 | grep -v amdgpu.ids | grep dense | sed "s/^/rows256=$R: /"; done
