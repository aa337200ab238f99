#!/bin/bash
export RSA_TUNING=1
python -m pytest tests/test_gpu_fp8.py -x -q -k "pv" -s 2>&1 | grep -v "amdgpu.ids" | tail -14
python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from bench import WORKLOADS, REGIMES, gen_inputs, make_neighbors, make_spec
from rectified_spaattn_amd import _core
from perf_k5 import timeit
dev = torch.device("cuda:0")
for wname, regime in (("hunyuan_720p_128f", "locality"), ("hunyuan_720p_128f", "r2"), ("wan22_ti2v_720p_121f", "r2")):
    wl = WORKLOADS[wname]; spec = make_spec(wl); cent, nbk, p = REGIMES[regime]
    q, k, v = gen_inputs(wl, wl["H"], 0, dev, cent)
    nbr = make_neighbors(wl, spec, nbk)
    res = {}
    for mode in (False, "pv", True):
        c = _core.StagedCall(q, k, v, spec, wl["top_k"], p, nbr, qkv_fp8=mode)
        c.select(); torch.cuda.synchronize()
        med, mn = timeit(c.attend, n=7, warm=2)
        msel, _ = timeit(c.select, n=5, warm=1)
        res[mode] = (med, msel, c.out.float().clone())
        del c
    ref = res[False][2]
    for mode in ("pv", True):
        d = (res[mode][2] - ref).abs()
        print(f"{wname} {regime}: {mode!s:5} K5 {res[mode][0]:.3f} ms select {res[mode][1]:.3f} | bf16 K5 {res[False][0]:.3f} | rel-L1 {float(d.sum() / ref.abs().sum()):.4f} max {float(d.max()):.3f} mean {float(d.mean()):.4f}", flush=True)
    del q, k, v, res, ref
    torch.cuda.empty_cache()
PY
