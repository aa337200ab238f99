#!/usr/bin/env python3
"""Per-kernel times of the mask-selection pass (K1..K4) at a bench workload, by events around each C-ABI call, interleaved over
the tuning keys given as key=a,b pairs (e.g. k2_v2=0,1).  RSA_PERF_H = heads (default the workload's), RSA_PERF_WORKLOAD."""
import ctypes
import os
import sys

os.environ.setdefault("RSA_TUNING", "1")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import REGIMES, WORKLOADS, gen_inputs, make_neighbors, make_spec, regime_top_k  # noqa: E402
from rectified_spaattn_amd import _core, _lib  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    wl = WORKLOADS[os.environ.get("RSA_PERF_WORKLOAD", "hunyuan_720p_128f")]
    H = int(os.environ.get("RSA_PERF_H", str(wl["H"])))
    regime = os.environ.get("RSA_PERF_REGIME", "r2")
    cent, nbr_kind, p = REGIMES[regime]
    spec = make_spec(wl)
    q, k, v = gen_inputs(wl, H, 0, dev, cent, D=wl.get("D", 128))
    call = _core.StagedCall(q, k, v, spec, regime_top_k(wl, regime), p, make_neighbors(wl, spec, nbr_kind))
    L = _lib.lib()
    lay, cb, st = ctypes.byref(call.lay), ctypes.byref(call.cb), _core._stream()
    tq, tk, tv = call.t
    nb = call.nbr.data_ptr() if call.nbr is not None else None
    stages = [("K1", lambda: L.rsa_pool_stats(lay, tq, tk, tv, cb, st)), ("K2", lambda: L.rsa_pooled_scores(lay, tk, cb, st)),
              ("K3", lambda: L.rsa_select_mask(lay, nb, call.top_k, call.p, cb, st)), ("K4", lambda: L.rsa_compensation(lay, cb, st))]
    variants = [{}]
    for a in sys.argv[1:]:
        key, vals = a.split("=")
        variants = [dict(vv, **{key: int(x)}) for vv in variants for x in vals.split(",")]
    res = {i: {n: [] for n, _ in stages + [("pass", None)]} for i in range(len(variants))}
    for rnd in range(8):
        for i, var in enumerate(variants):
            for key, val in var.items():
                assert L.rsa_set_tuning(key.encode(), val) == 0, key
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(stages) + 1)]
            torch.cuda.synchronize()
            evs[0].record()
            for j, (_, fn) in enumerate(stages):
                assert fn() == 0
                evs[j + 1].record()
            torch.cuda.synchronize()
            if rnd >= 2:
                for j, (n, _) in enumerate(stages):
                    res[i][n].append(evs[j].elapsed_time(evs[j + 1]) * 1e3)
                res[i]["pass"].append(evs[0].elapsed_time(evs[-1]) * 1e3)
    for i, var in enumerate(variants):
        med = {n: sorted(x)[len(x) // 2] for n, x in res[i].items()}
        print(f"{wl['variant']} H={H} {var or 'default'}: " + " | ".join(f"{n} {med[n]:.1f} us" for n in med), flush=True)


if __name__ == "__main__":
    main()
