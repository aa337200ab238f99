#!/bin/bash
# round 6: the sub-step boundary (vmcnt wait, barrier, the next block's first K reads) two / four MFMAs earlier -- head dim 128 as forms of the
# A/B build (27 = static body as it is, 32 / 33 = boundary in front of MFMA 28 / 26; 0 = product, 34 = online body with the boundary at 28),
# head dim 64 as whole builds (CogVideoX bench line per build, two rounds)
mkdir -p gpurun_out
export RSA_TUNING=1
FORMS="0 27 32 33 34" ROUNDS=6 bash tools/r6_forms.sh r6f5
P=rectified_spaattn_amd
cp $P/librsa_hip.so /tmp/librsa_hip_product.so
for R in 1 2; do
for V in product c4 c6; do
  if [ $V = product ]; then cp /tmp/librsa_hip_product.so $P/librsa_hip.so; else cp $P/librsa_hip_$V.so $P/librsa_hip.so; fi
  python bench.py --steps 20 --warmup 3 --workload cogvideox_768p_81f --no-cpu-baseline --no-extras --no-live-traffic > gpurun_out/r6o_$V.json 2>> gpurun_out/r6o.err
  python -c "import json;d=json.load(open('gpurun_out/r6o_$V.json'));print('$V round $R',d['ms_per_step'],d['roofline']['frac'],d['roofline']['k5_ms'],d['check']['ok'])"
done
done 2>&1 | tee gpurun_out/r6o_variants.txt
cp /tmp/librsa_hip_product.so $P/librsa_hip.so
