#!/bin/bash
# times the pv and the e4m3 K5 in each placement library of tools/history/r5_gaps_build.sh (one process per library and form)

for rep in 1 2; do
for x in ${RSA_GAPS_LIBS:-base early even2 late pvphase}; do
python - $x <<'PY' 2>&1 | grep -v amdgpu.ids
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
    for mode in ("pv", True):
        c = _core.StagedCall(q, k, v, spec, wl["top_k"], p, make_neighbors(wl, spec, nbk), qkv_fp8=mode)
        c.select(); torch.cuda.synchronize()
        med, mn = timeit(c.attend, n=9, warm=3)
        out.append(f"{regime} {mode}: {med:.3f} (min {mn:.3f})")
        del c
    del q, k, v
print(f"{name:8s} K5 ms: " + " | ".join(out), flush=True)
PY
done
done
