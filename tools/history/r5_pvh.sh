#!/bin/bash
# hand-placed block of the pv form (fp8_variant 0) against its compiled twin (fp8_variant 1): tests, bit equality, A/B timing
export RSA_TUNING=1
python -m pytest tests/test_gpu_fp8.py -x -q ${RSA_PVH_TESTS:--k pv} 2>&1 | grep -v "amdgpu.ids" | tail -6
python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from bench import WORKLOADS, REGIMES, gen_inputs, make_neighbors, make_spec
from rectified_spaattn_amd import _core, _lib
from perf_k5 import timeit
L = _lib.lib()
dev = torch.device("cuda:0")
MODE = "pv" if os.environ.get("RSA_PVH_MODE", "pv") == "pv" else True
VARS = [int(x) for x in os.environ.get("RSA_PVH_VARS", "1,0").split(",")]
NAMES = {0: "product", 1: "compiled", 2: "exact-exp", 3: "hand-placed, staging behind the barrier"}
for wname, regime in (("hunyuan_720p_128f", "r2"), ("hunyuan_720p_128f", "locality"), ("wan22_ti2v_720p_121f", "r2")):
    wl = WORKLOADS[wname]; spec = make_spec(wl); cent, nbk, p = REGIMES[regime]
    q, k, v = gen_inputs(wl, wl["H"], 0, dev, cent)
    nbr = make_neighbors(wl, spec, nbk)
    cb = _core.StagedCall(q, k, v, spec, wl["top_k"], p, nbr, qkv_fp8=False)
    cb.select(); torch.cuda.synchronize()
    medb, _ = timeit(cb.attend, n=5, warm=2)
    ref = cb.out.float().clone()
    del cb
    c = _core.StagedCall(q, k, v, spec, wl["top_k"], p, nbr, qkv_fp8=MODE)
    c.select(); torch.cuda.synchronize()
    outs = {}
    for rnd in range(2):
        for var in VARS:
            assert L.rsa_set_tuning(b"fp8_variant", var) == 0
            med, mn = timeit(c.attend, n=7, warm=2)
            outs[var] = c.out.float().clone()
            d = (outs[var] - ref).abs()
            print(f"{wname} {regime} round {rnd}: {MODE} variant {var} ({NAMES[var]}) K5 {med:.3f} ms (min {mn:.3f}) | bf16 {medb:.3f} | rel-L1 {float(d.sum() / ref.abs().sum()):.4f} max {float(d.max()):.3f}", flush=True)
    L.rsa_set_tuning(b"fp8_variant", 0)
    dd = (outs[VARS[0]] - outs[VARS[-1]]).abs()
    print(f"  first vs last variant: max |d| {float(dd.max()):.3e}, differing elements {int((dd > 0).sum())} of {dd.numel()}", flush=True)
    del q, k, v, c, outs, ref
    torch.cuda.empty_cache()
PY
