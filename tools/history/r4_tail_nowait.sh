export RSA_TUNING=1
L=rectified_spaattn_amd/librsa_hip.so
for H in 3 4; do
  RSA_PERF_H=$H timeout 600 python tools/ab_libs.py whole=$L::k5_w64=1,k5_tail_split=0 split=$L::k5_w64=1,k5_tail_split=1 --rounds 10 2>&1 | grep -E "sparse median" | cut -c1-110 | sed "s/^/heads $H /"
done
for G in 0 1 0 1; do RSA_K5_TAIL_SPLIT=$G python bench.py --steps 20 --warmup 3 --workload wan22_ti2v_720p_121f --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('wan22 tail_split=$G', r['ms_per_step'], r['roofline']['frac'], r['roofline']['k5_ms'])"; done
timeout 600 python -m pytest tests/test_gpu_tail_split.py tests/test_gpu_gsync.py -q -m gpu 2>&1 | grep -E "passed|failed"
