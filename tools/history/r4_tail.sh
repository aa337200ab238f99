#!/bin/bash
# round 4: tail split of the 64-row K5: the whole GPU suite with it on (default), per-shape K5 times with the switch off / on
export RSA_TUNING=1
mkdir -p gpurun_out
T=${1:-r4t}
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -8 ) > gpurun_out/${T}_tests.txt 2>&1
L=rectified_spaattn_amd/librsa_hip.so
for H in 24 12 6 3; do
  ( RSA_PERF_H=$H timeout 600 python tools/ab_libs.py whole=$L::k5_w64=1,k5_tail_split=0 split=$L::k5_w64=1,k5_tail_split=1 --rounds 8 ) > gpurun_out/${T}_ab_h$H.txt 2>&1
  echo "heads $H"; tail -2 gpurun_out/${T}_ab_h$H.txt | cut -c1-150; grep "max|out" gpurun_out/${T}_ab_h$H.txt
done
for WL in wan22_ti2v_720p_121f flux_4096 wan21_720p_81f; do
  for G in 0 1 0 1; do
    RSA_K5_TAIL_SPLIT=$G python bench.py --steps 20 --warmup 3 --workload $WL --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" >> gpurun_out/${T}_bench_${WL}_s$G.json
  done
done
cat gpurun_out/${T}_tests.txt
python3 - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${T}_bench_*.json")):
    for line in open(f).read().strip().splitlines():
        r=json.loads(line); print(f.split("bench_")[1], r["ms_per_step"], r["roofline"]["frac"], r["roofline"].get("k5_ms"))
PY
