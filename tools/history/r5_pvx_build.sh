#!/bin/bash
# builds the timing-experiment libraries of the pv form's tile block (run HERE, the .so files travel with gpurun):
#   rectified_spaattn_amd/librsa_hip_x_<name>.so, name in: base halfk nov nolds novalu nodma nobar nodmabar (pv block) e8halfk e8nolds (e4m3 block)
# (experiments drop one resource each; results are garbage, only the time means something).  tools/history/r5_pvx.sh runs them.
set -e
cd "$(dirname "$0")/../../rectified_spaattn_amd/csrc"
make -s
OBJS="rsa_stats.o rsa_attn.o rsa_attn_kernel.o rsa_attn_kernel64.o rsa_attn_masked.o rsa_fp8.o rsa_glue.o rsa_geometry.o rsa_comm.o"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -I. -Wall -Wno-unused-function -fno-honor-nans"
build() {   # name, generator flags of the pv block, compiler defines, generator flags of the e4m3 block
    RSA_GEN8H_X="$2" RSA_GEN8_X="$4" python3 gen_k5_block.py > rsa_attn_block.h
    /opt/rocm/bin/hipcc $FLAGS $3 -c rsa_attn_fp8_kernel.hip -o /tmp/rsa_attn_fp8_kernel.x_$1.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librsa_hip_x_$1.so $OBJS /tmp/rsa_attn_fp8_kernel.x_$1.o -ldl
    echo "built librsa_hip_x_$1.so"
}
build base "" ""
build halfk halfk ""
build nov nov ""
build nolds nok,nov ""
build novalu novalu ""
build nodma "" -DRSA_PVX_NODMA
build nobar "" -DRSA_PVX_NOBAR
build nodmabar "" "-DRSA_PVX_NODMA -DRSA_PVX_NOBAR"
build hotdma "" -DRSA_PVX_HOTDMA
build e8halfk "" "" halfk
build e8nolds "" "" nolds
python3 gen_k5_block.py > rsa_attn_block.h     # back to the product's header
touch rsa_attn_block.h
make -s
