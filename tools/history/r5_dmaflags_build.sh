#!/bin/bash
# cache-policy bits on the LDS-DMA loads of the pv / e4m3 blocks: one library per choice (built here), timed by r5_gaps.sh-style loop
set -e
cd "$(dirname "$0")/../../rectified_spaattn_amd/csrc"
make -s
OBJS="rsa_stats.o rsa_attn.o rsa_attn_kernel.o rsa_attn_kernel64.o rsa_attn_masked.o rsa_fp8.o rsa_glue.o rsa_geometry.o rsa_comm.o"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -I. -Wall -Wno-unused-function -fno-honor-nans"
build() {
    RSA_GEN_DMAFLAGS="$2" python3 gen_k5_block.py > rsa_attn_block.h
    /opt/rocm/bin/hipcc $FLAGS -c rsa_attn_fp8_kernel.hip -o /tmp/rsa_attn_fp8_kernel.x_$1.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librsa_hip_x_$1.so $OBJS /tmp/rsa_attn_fp8_kernel.x_$1.o -ldl
    echo "built librsa_hip_x_$1.so"
}
build base ""
# (needs the RSA_GEN_DMAFLAGS hook in gen_k5_block.py: see the commit that added this script)
build sc0 " sc0"
build sc1 " sc1"
build sc01 " sc0 sc1"
build nt " nt"
python3 gen_k5_block.py > rsa_attn_block.h
touch rsa_attn_block.h
make -s
