#!/bin/bash
# aligned starts (tuning key k5_gsync bit 1) for the pv / e4m3 kernels after the staging moved into the block: A/B in one process
export RSA_TUNING=1
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from bench import WORKLOADS, REGIMES, gen_inputs, make_neighbors, make_spec, regime_top_k
from rectified_spaattn_amd import _core, _lib
from perf_k5 import timeit
L = _lib.lib()
dev = torch.device("cuda:0")
for wname, regime in (("hunyuan_720p_128f", "r2"), ("hunyuan_720p_128f", "script"), ("wan21_720p_81f", "script"), ("wan22_ti2v_720p_121f", "r2")):
    wl = WORKLOADS[wname]; spec = make_spec(wl); cent, nbk, p = REGIMES[regime]
    q, k, v = gen_inputs(wl, wl["H"], 0, dev, cent)
    nbr = make_neighbors(wl, spec, nbk)
    for mode in ("pv", True):
        c = _core.StagedCall(q, k, v, spec, regime_top_k(wl, regime), p, nbr, qkv_fp8=mode)
        c.select(); torch.cuda.synchronize()
        res = []
        for rnd in range(2):
            for gs in (1, 3):
                assert L.rsa_set_tuning(b"k5_gsync", gs) == 0
                med, mn = timeit(c.attend, n=7, warm=2)
                res.append(f"gsync {gs}: {med:.3f} (min {mn:.3f})")
        L.rsa_set_tuning(b"k5_gsync", 1)
        print(f"{wname} {regime} {mode}: " + " | ".join(res), flush=True)
        del c
    del q, k, v
    torch.cuda.empty_cache()
PY
