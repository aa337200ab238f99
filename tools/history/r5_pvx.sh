#!/bin/bash
# times the pv form's K5 in each experiment library (tools/history/r5_pvx_build.sh) on the R2 / locality regimes, one process per library
# usage: tools/history/r5_pvx.sh [pv|e4m3]
MODE=${1:-pv}
if [ "$MODE" = pv ]; then LIBS="${RSA_PVX_LIBS:-base halfk nov nolds novalu nodma nobar nodmabar hotdma base}"; else LIBS="${RSA_PVX_LIBS:-base e8halfk e8nolds nodma nobar nodmabar hotdma base}"; fi
for x in $LIBS; do
python - $x $MODE <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from rectified_spaattn_amd import _lib
name = sys.argv[1]
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.__file__), f"librsa_hip_x_{name}.so")
from bench import WORKLOADS, REGIMES, gen_inputs, make_neighbors, make_spec
from rectified_spaattn_amd import _core
from perf_k5 import timeit
dev = torch.device("cuda:0")
wl = WORKLOADS["hunyuan_720p_128f"]; spec = make_spec(wl)
out = []
for regime in ("r2", "locality"):
    cent, nbk, p = REGIMES[regime]
    q, k, v = gen_inputs(wl, wl["H"], 0, dev, cent)
    c = _core.StagedCall(q, k, v, spec, wl["top_k"], p, make_neighbors(wl, spec, nbk), qkv_fp8="pv" if sys.argv[2] == "pv" else True)
    c.select(); torch.cuda.synchronize()
    med, mn = timeit(c.attend, n=9, warm=3)
    out.append(f"{regime} {med:.3f} (min {mn:.3f})")
    del c, q, k, v
print(f"{sys.argv[2]} {name:9s} K5 ms: " + " | ".join(out), flush=True)
PY
done
