#!/bin/bash
# head dim 64 (CogVideoX): row sums on the matrix pipe -- tests of the D = 64 paths + K5 timing (product library)
python -m pytest tests -x -q -m gpu -k "cogvideo or d64 or head_dim_64 or dim64 or D64 or 64" 2>&1 | grep -v amdgpu.ids | tail -5
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from bench import WORKLOADS, REGIMES, gen_inputs, make_neighbors, make_spec, regime_top_k
from rectified_spaattn_amd import _core
from perf_k5 import timeit
dev = torch.device("cuda:0")
wl = WORKLOADS["cogvideox_768p_81f"]; spec = make_spec(wl)
for regime in ("r2", "locality", "script"):
    cent, nbk, p = REGIMES[regime]
    q, k, v = gen_inputs(wl, wl["H"], 0, dev, cent, D=64)
    c = _core.StagedCall(q, k, v, spec, regime_top_k(wl, regime), p, make_neighbors(wl, spec, nbk))
    c.select(); torch.cuda.synchronize()
    pairs = c.bufs["counts"].sum().item()
    fl = 4.0 * 64 * 128 * 128 * pairs + 4.0 * 64 * spec.q_text_valid * spec.kv_text_valid * wl["H"]
    med, mn = timeit(c.attend, n=9, warm=3)
    print(f"D=64 {regime}: K5 {med:.3f} ms (min {mn:.3f}) {fl/med/1e9:.0f} TFLOP/s = {fl/med/1e9/2500:.3f} of 2.5 PF", flush=True)
    del c, q, k, v
PY
