#!/bin/bash
# round 6: timing-only forms of the 64-row loop (librsa_hip_ab.so, built here): x14 = every 32x32x16 MFMA as two 16x16x32 (the
# guide's DVFS item 7: does the chip hold a higher clock on that shape?), x15 = every second LDS-DMA piece dropped (what a 256-row
# workgroup sharing one K/V ring would issue per wave), x16 = both, x5 / x17 = MFMAs + LDS operand reads only on the two shapes.
export RSA_TUNING=1
mkdir -p gpurun_out
T=${1:-r6f}
make -C rectified_spaattn_amd/csrc ab > gpurun_out/${T}_make.txt 2>&1 || { tail -20 gpurun_out/${T}_make.txt; exit 1; }
L=rectified_spaattn_amd/librsa_hip_ab.so
S=""
for n in ${FORMS:-0 14 15 16 5 17}; do S="$S x$n=$L::k5w_form=$n"; done
( timeout 900 python tools/ab_libs.py $S --rounds ${ROUNDS:-6} ) > gpurun_out/${T}_forms.txt 2>&1
echo "rc=$?" >> gpurun_out/${T}_forms.txt
tail -24 gpurun_out/${T}_forms.txt | cut -c1-260
