#!/bin/bash
# round 4: aligned starts in every K5 kernel: whole GPU suite + smoke, A/B of the 2-byte and the e4m3 kernel (one process each),
# head dim 64 and the other workloads with the switch off / on, two ranks on one device (the case the bounded wait is for)
export RSA_TUNING=1
mkdir -p gpurun_out
T=${1:-r4j}
L=rectified_spaattn_amd/librsa_hip.so
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -6 ) > gpurun_out/${T}_tests.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) > gpurun_out/${T}_smoke.txt
( timeout 600 python tools/ab_libs.py free=$L::k5_w64=1,k5_gsync=0 aligned=$L::k5_w64=1,k5_gsync=1 free32=$L::k5_w64=0,k5_gsync=0 aligned32=$L::k5_w64=0,k5_gsync=1 --rounds 6 ) > gpurun_out/${T}_ab.txt 2>&1
( timeout 600 python tools/ab_libs.py free=$L::k5_gsync=0 aligned=$L::k5_gsync=1 --rounds 6 --fp8 ) > gpurun_out/${T}_ab_fp8.txt 2>&1
for WL in cogvideox_768p_81f flux_4096 wan21_720p_81f wan22_ti2v_720p_121f; do
  for G in 0 1 0 1; do
    RSA_K5_GSYNC=$G python bench.py --steps 20 --warmup 3 --workload $WL --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" >> gpurun_out/${T}_bench_${WL}_g$G.json
    RSA_K5_GSYNC=$G python bench.py --steps 20 --warmup 3 --workload $WL --no-cpu-baseline --no-extras --qkv-fp8 2>/dev/null | grep "^{" >> gpurun_out/${T}_bench_${WL}_fp8_g$G.json
  done
done
( RSA_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-extras --no-cpu-baseline | grep "^{" ) > gpurun_out/${T}_bench_2ranks.json 2> gpurun_out/${T}_bench_2ranks.err
cat gpurun_out/${T}_tests.txt gpurun_out/${T}_smoke.txt
tail -5 gpurun_out/${T}_ab.txt | cut -c1-200; tail -3 gpurun_out/${T}_ab_fp8.txt | cut -c1-200
python3 - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${T}_bench_*.json")):
    for line in open(f).read().strip().splitlines():
        try:
            r=json.loads(line); print(f.split("bench_")[1], r["ms_per_step"], r["roofline"]["frac"], r["roofline"].get("k5_ms"))
        except Exception as e: print(f, "ERR", e)
PY
