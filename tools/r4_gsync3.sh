#!/bin/bash
# round 4: aligned starts of the 64-row K5: all-of-64 / quorum forms against free-running walks and the 32-row kernel, per regime
# (every spec sets BOTH keys: the builds share one library handle, a key left alone keeps the previous spec's value)
export RSA_TUNING=1
mkdir -p gpurun_out
T=${1:-r4i}
L=rectified_spaattn_amd/librsa_hip.so
for RG in r2 r1 locality; do
  ( RSA_PERF_REGIME=$RG timeout 600 python tools/ab_libs.py free=$L::k5_w64=1,k5_gsync=0 aligned=$L::k5_w64=1,k5_gsync=1 q56=$L::k5_w64=1,k5_gsync=56 q48=$L::k5_w64=1,k5_gsync=48 q32=$L::k5_w64=1,k5_gsync=32 row32=$L::k5_w64=0,k5_gsync=0 --rounds 6 ) > gpurun_out/${T}_ab_$RG.txt 2>&1
  echo $RG; tail -7 gpurun_out/${T}_ab_$RG.txt | cut -c1-220
done
for WL in flux_4096 wan21_720p_81f wan22_ti2v_720p_121f; do
  for G in 0 1 48; do
    RSA_K5_GSYNC=$G python bench.py --steps 20 --warmup 3 --workload $WL --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" > gpurun_out/${T}_bench_${WL}_g$G.json
  done
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${T}_bench_*_g*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); print(f, r["ms_per_step"], r["roofline"]["frac"], r["roofline"].get("k5_ms"))
    except Exception as e: print(f, "ERR", e)
PY
