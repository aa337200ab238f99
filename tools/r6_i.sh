#!/bin/bash
# round 6: schedule variants of the head-dim-64 64-row loop (vector work per MFMA gap 30 / 36 / 44 issue cycles, 3 / 4 exponentials per gap):
# the CogVideoX bench line per build on ONE box, two rounds; then the stamps of the head-dim-128 kernel's STATIC body (diag build)
mkdir -p gpurun_out
export RSA_TUNING=1
P=rectified_spaattn_amd
cp $P/librsa_hip.so /tmp/librsa_hip_product.so
for R in 1 2; do
for V in product g30_3 g44_3 g36_4; do
  if [ $V = product ]; then cp /tmp/librsa_hip_product.so $P/librsa_hip.so; else cp $P/librsa_hip_$V.so $P/librsa_hip.so; fi
  python bench.py --steps 20 --warmup 3 --workload cogvideox_768p_81f --no-cpu-baseline --no-extras --no-live-traffic > gpurun_out/r6i_$V.json 2>> gpurun_out/r6i.err
  python -c "import json;d=json.load(open('gpurun_out/r6i_$V.json'));print('$V round $R',d['ms_per_step'],d['roofline']['frac'],d['roofline']['k5_ms'],d['check']['ok'])"
done
done 2>&1 | tee gpurun_out/r6i_variants.txt
cp /tmp/librsa_hip_product.so $P/librsa_hip.so
make -s -C $P/csrc diag > /dev/null 2>&1
python tools/diag_k5w.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6i_diag.txt; cat gpurun_out/r6i_diag.txt | cut -c1-700
