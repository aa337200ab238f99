#!/bin/bash
# round 4: what each part of the 64-row loop costs: the A/B forms of librsa_hip_ab.so (work removed / re-priced), interleaved in
# one process on one device.  Forms 1-6 and 8 compute garbage by design; only the times are read.
export RSA_TUNING=1
mkdir -p gpurun_out
T=${1:-r4f}
L=rectified_spaattn_amd/librsa_hip_ab.so
S=""
for n in ${FORMS:-0 1 2 3 4 5 6 7 8}; do S="$S x$n=$L::k5w_form=$n"; done
( timeout 900 python tools/ab_libs.py $S --rounds ${ROUNDS:-6} ) > gpurun_out/${T}_forms.txt 2>&1
echo "rc=$?" >> gpurun_out/${T}_forms.txt
tail -24 gpurun_out/${T}_forms.txt | cut -c1-260
