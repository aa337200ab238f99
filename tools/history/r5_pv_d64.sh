#!/bin/bash
# the pv form at head dim 64 (CogVideoX 768p): K5 of the 2-byte / pv (product, compiled twin) / e4m3 (product, variant 3 = staging
# behind the barrier + ones read from LDS) kernels + accuracy, one process
export RSA_TUNING=1
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from bench import WORKLOADS, REGIMES, gen_inputs, make_neighbors, make_spec, regime_top_k
from rectified_spaattn_amd import _core, _lib
from perf_k5 import timeit
L = _lib.lib()
dev = torch.device("cuda:0")
wl = WORKLOADS["cogvideox_768p_81f"]; spec = make_spec(wl)
for regime in ("r2", "script"):
    cent, nbk, p = REGIMES[regime]
    q, k, v = gen_inputs(wl, wl["H"], 0, dev, cent, D=64)
    nbr = make_neighbors(wl, spec, nbk)
    res = {}
    for mode in (False, "pv", True):
        c = _core.StagedCall(q, k, v, spec, regime_top_k(wl, regime), p, nbr, qkv_fp8=mode)
        c.select(); torch.cuda.synchronize()
        pairs = c.bufs["counts"].sum().item()
        for var in ((0, 1) if mode == "pv" else ((0, 3) if mode is True else (0,))):
            L.rsa_set_tuning(b"fp8_variant", var)
            med, mn = timeit(c.attend, n=9, warm=3)
            res[(mode, var)] = (med, c.out.float().clone())
        L.rsa_set_tuning(b"fp8_variant", 0)
        msel, _ = timeit(c.select, n=5, warm=1)
        res[(mode, "sel")] = msel
        del c
    ref = res[(False, 0)][1]
    fl = 4.0 * 64 * 128 * 128 * pairs + 4.0 * 64 * spec.q_text_valid * spec.kv_text_valid * wl["H"]
    for key in ((False, 0), ("pv", 0), ("pv", 1), (True, 0), (True, 3)):
        med, o = res[key]
        d = (o - ref).abs()
        print(f"CogVideoX {regime}: {str(key[0]):5} variant {key[1]}: K5 {med:.3f} ms = {fl/med/1e9:.0f} TFLOP/s, select {res[(key[0], 'sel')]:.3f} ms | vs 2-byte: rel-L1 {float(d.sum()/ref.abs().sum()):.4f} max {float(d.max()):.3f}", flush=True)
    del q, k, v, res, ref
    torch.cuda.empty_cache()
PY
