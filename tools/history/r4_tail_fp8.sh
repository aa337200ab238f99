#!/bin/bash
# tail split in the e4m3 K5: tests, Wan2.2-TI2V (config 5) and the headline workload with the switch off / on, 3 heads
export RSA_TUNING=1
timeout 900 python -m pytest tests/test_gpu_tail_split.py tests/test_gpu_fp8.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for G in 0 1 0 1; do RSA_K5_TAIL_SPLIT=$G python bench.py --steps 20 --warmup 3 --workload wan22_ti2v_720p_121f --qkv-fp8 --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('wan22 e4m3 tail_split=$G', r['ms_per_step'], r['roofline']['frac'], r['roofline']['k5_ms'])"; done
L=rectified_spaattn_amd/librsa_hip.so
for H in 24 3; do RSA_PERF_H=$H timeout 600 python tools/ab_libs.py whole=$L::k5_tail_split=0 split=$L::k5_tail_split=1 --rounds 8 --fp8 2>&1 | grep -E "sparse median|max\|out" | cut -c1-110 | sed "s/^/e4m3 heads $H /"; done
