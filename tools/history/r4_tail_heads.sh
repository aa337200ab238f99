export RSA_TUNING=1
L=rectified_spaattn_amd/librsa_hip.so
for H in 3 4 5 2; do
  RSA_PERF_H=$H timeout 600 python tools/ab_libs.py whole=$L::k5_w64=1,k5_tail_split=0 split=$L::k5_w64=1,k5_tail_split=1 --rounds 8 2>&1 | grep -E "sparse median|max\|out" | cut -c1-110 | sed "s/^/heads $H /"
done
timeout 600 python -m pytest tests/test_gpu_tail_split.py -q -m gpu 2>&1 | grep -E "passed|failed"
