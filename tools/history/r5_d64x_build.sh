#!/bin/bash
# timing-experiment libraries of the 2-byte 32-row kernel (head dim 64 = CogVideoX is what they are for): one resource removed each
#   rectified_spaattn_amd/librsa_hip_x_<name>.so, name in: base noexp novalu nolds nodma nobar        (tools/history/r5_d64x.sh runs them)
set -e
cd "$(dirname "$0")/../../rectified_spaattn_amd/csrc"
make -s
OBJS="rsa_stats.o rsa_attn.o rsa_attn_kernel64.o rsa_attn_masked.o rsa_fp8.o rsa_attn_fp8_kernel.o rsa_glue.o rsa_geometry.o rsa_comm.o"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -I. -Wall -Wno-unused-function -fno-honor-nans"
build() {   # name, generator flags, compiler defines, (4th: non-empty = row sums by vector additions)
    RSA_GEN_X="$2" RSA_GEN_NORSM="$4" python3 gen_k5_block.py > rsa_attn_block.h
    /opt/rocm/bin/hipcc $FLAGS $3 -c rsa_attn_kernel.hip -o /tmp/rsa_attn_kernel.x_$1.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librsa_hip_x_$1.so $OBJS /tmp/rsa_attn_kernel.x_$1.o -ldl
    echo "built librsa_hip_x_$1.so"
}
build base "" ""
build noexp noexp ""
build novalu novalu ""
build nolds nolds ""
build nodma "" -DRSA_K5X_NODMA
build nobar "" -DRSA_K5X_NOBAR
build norsm "" -DRSA_K5X_NORSM 1
python3 gen_k5_block.py > rsa_attn_block.h     # back to the product's header
touch rsa_attn_block.h
make -s
