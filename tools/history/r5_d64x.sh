#!/bin/bash
# times K5 (2-byte, head dim 64: CogVideoX workload) in each experiment library of tools/history/r5_d64x_build.sh, one process per library
for x in ${RSA_D64X_LIBS:-base noexp novalu nolds nodma nobar norsm base}; do
python - $x <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from rectified_spaattn_amd import _lib
name = sys.argv[1]
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.__file__), f"librsa_hip_x_{name}.so")
from bench import WORKLOADS, REGIMES, gen_inputs, make_neighbors, make_spec, regime_top_k
from rectified_spaattn_amd import _core
from perf_k5 import timeit
dev = torch.device("cuda:0")
wl = WORKLOADS["cogvideox_768p_81f"]; spec = make_spec(wl)
out = []
for regime in ("r2", "locality"):
    cent, nbk, p = REGIMES[regime]
    q, k, v = gen_inputs(wl, wl["H"], 0, dev, cent, D=wl.get("D", 128))
    c = _core.StagedCall(q, k, v, spec, regime_top_k(wl, regime), p, make_neighbors(wl, spec, nbk))
    c.select(); torch.cuda.synchronize()
    pairs = c.bufs["counts"].sum().item()
    med, mn = timeit(c.attend, n=9, warm=3)
    out.append(f"{regime} {med:.3f} (min {mn:.3f}) pairs {pairs}")
    del c, q, k, v
print(f"D=64 {name:7s} K5 ms: " + " | ".join(out), flush=True)
PY
done
