#!/bin/bash
# round 6: the 64-row K5 at head dim 64: every test that touches head dim 64, then the CogVideoX bench line with the 64-row and the 32-row kernel,
# then the in-kernel stamps of the head-dim-128 kernel (diag build with the stamp register fixed)
mkdir -p gpurun_out
export RSA_TUNING=1
( timeout 1500 python -m pytest tests/test_gpu_select_paths.py tests/test_gpu_parity.py tests/test_gpu_gsync.py tests/test_gpu_random_layouts.py tests/test_gpu_fullsize.py tests/test_gpu_static_reference.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -30 ) > gpurun_out/r6h_tests.txt 2>&1
tail -12 gpurun_out/r6h_tests.txt
for W in 3 1; do
  RSA_K5_W64=$W python bench.py --steps 20 --warmup 3 --workload cogvideox_768p_81f --no-cpu-baseline --no-extras --no-live-traffic > gpurun_out/r6h_bench_cogvideox_w$W.json 2>> gpurun_out/r6h_bench.err
  python -c "import json;d=json.load(open('gpurun_out/r6h_bench_cogvideox_w$W.json'));print('w64=$W',d['ms_per_step'],d['value'],d['roofline']['frac'],d['roofline']['k5_ms'],d['check']['ok'],d['check']['max_abs'])"
done
make -s -C rectified_spaattn_amd/csrc diag > /dev/null 2>&1
python tools/diag_k5w.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6h_diag.txt; cat gpurun_out/r6h_diag.txt | cut -c1-400
