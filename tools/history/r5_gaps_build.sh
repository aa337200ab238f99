#!/bin/bash
# placement A/B of the LDS-DMA pieces inside the pv / e4m3 blocks: librsa_hip_x_<name>.so per placement (built here)
set -e
cd "$(dirname "$0")/../../rectified_spaattn_amd/csrc"
make -s
OBJS="rsa_stats.o rsa_attn.o rsa_attn_kernel.o rsa_attn_kernel64.o rsa_attn_masked.o rsa_fp8.o rsa_glue.o rsa_geometry.o rsa_comm.o"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -I. -Wall -Wno-unused-function -fno-honor-nans"
build() {   # name, pv gaps, e4m3 gaps
    RSA_GEN8H_GAPS="$2" RSA_GEN8_GAPS="$3" python3 gen_k5_block.py > rsa_attn_block.h
    /opt/rocm/bin/hipcc $FLAGS -c rsa_attn_fp8_kernel.hip -o /tmp/rsa_attn_fp8_kernel.x_$1.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librsa_hip_x_$1.so $OBJS /tmp/rsa_attn_fp8_kernel.x_$1.o -ldl
    echo "built librsa_hip_x_$1.so"
}
build base "" ""
build early 0,1,2,3,4,5 0,1,2,3
build even2 0,2,4,6,8,10 1,3,5,7
build late 8,10,12,14,16,18 4,5,6,7
build pvphase 15,16,17,18,19,20 5,6,7,8
python3 gen_k5_block.py > rsa_attn_block.h
touch rsa_attn_block.h
make -s
