#!/bin/bash
# hand-placed block: tests first (bit-identity with the compiled block is part of them), then A/B in one process
R=$PWD; TAG=${1:-r3b}; mkdir -p $R/gpurun_out
python3 tools/check_blk.py > $R/gpurun_out/${TAG}_check.txt 2>&1; tail -12 $R/gpurun_out/${TAG}_check.txt
C=$R/rectified_spaattn_amd/librsa_hip_ab.so
python3 tools/ab_libs.py r1=$R/build/librsa_hip_r1.so:14 form0=$C:15:k5_form=0 form1=$C:15:k5_form=1 prod=$R/rectified_spaattn_amd/librsa_hip.so:15 --rounds ${ROUNDS:-8} > $R/gpurun_out/${TAG}_ab.txt 2>&1
tail -6 $R/gpurun_out/${TAG}_ab.txt
python3 tools/diag_k5.py > $R/gpurun_out/${TAG}_diag.txt 2>&1; grep k5_form $R/gpurun_out/${TAG}_diag.txt
