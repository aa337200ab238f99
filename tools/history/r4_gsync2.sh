#!/bin/bash
# round 4: aligned starts (k5_gsync) as the product default: parity subset, A/B per regime, stamps
export RSA_TUNING=1
mkdir -p gpurun_out
T=${1:-r4h}
L=rectified_spaattn_amd/librsa_hip.so
( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/${T}_smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/${T}_smoke.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_fullsize.py tests/test_gpu_shard_invariance.py -x -q -m gpu ) > gpurun_out/${T}_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/${T}_tests.txt
for RG in r2 r1 locality; do
  ( RSA_PERF_REGIME=$RG timeout 600 python tools/ab_libs.py free=$L::k5_gsync=0 aligned=$L::k5_gsync=1 row32=$L::k5_w64=0 --rounds 6 ) > gpurun_out/${T}_ab_$RG.txt 2>&1
done
( timeout 300 python tools/diag_k5w.py ) > gpurun_out/${T}_diag.txt 2>&1
for WL in flux_4096 wan21_720p_81f wan22_ti2v_720p_121f; do
  for G in 0 1; do
    RSA_K5_GSYNC=$G python bench.py --steps 20 --warmup 3 --workload $WL --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" > gpurun_out/${T}_bench_${WL}_g$G.json
  done
done
tail -2 gpurun_out/${T}_smoke.txt; tail -3 gpurun_out/${T}_tests.txt | cut -c1-200
for RG in r2 r1 locality; do echo $RG; tail -3 gpurun_out/${T}_ab_$RG.txt | cut -c1-220; done
tail -2 gpurun_out/${T}_diag.txt
python3 - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${T}_bench_*_g*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); print(f, r["ms_per_step"], r["roofline"]["frac"], r["roofline"].get("k5_ms"))
    except Exception as e: print(f, "ERR", e)
PY
