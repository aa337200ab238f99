#!/bin/bash
# the split text-row pieces first / last in the grid of the e4m3 and the 32-row K5 (k5_text_last), interleaved; parity subsets
export RSA_TUNING=1
L=rectified_spaattn_amd/librsa_hip.so
timeout 600 python tools/ab_libs.py first=$L::k5_text_last=0 last=$L::k5_text_last=1 --rounds 10 --fp8 2>&1 | grep -E "sparse median|max\|out" | cut -c1-110 | sed 's/^/e4m3 /'
timeout 600 python tools/ab_libs.py first=$L::k5_w64=0,k5_text_last=0 last=$L::k5_w64=0,k5_text_last=1 --rounds 8 2>&1 | grep -E "sparse median|max\|out" | cut -c1-110 | sed 's/^/32-row /'
for G in 0 1 0 1; do RSA_K5_TEXT_LAST=$G python bench.py --steps 20 --warmup 3 --workload cogvideox_768p_81f --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('cogvideox text_last=$G', r['ms_per_step'], r['roofline']['frac'], r['roofline']['k5_ms'])"; done
timeout 900 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_parity.py tests/test_gpu_select_paths.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
