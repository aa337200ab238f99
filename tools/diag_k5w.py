#!/usr/bin/env python3
"""In-kernel stamps of the 64-rows-per-wave K5 (make diag -> librsa_hip_diag.so, tuning key k5_w64 = 1): cycles per 32-key
sub-step and wave inside the asm loop, in the C++-driven boundary blocks, and outside the walk; sparse R2 call."""
import ctypes
import os
import sys

os.environ["RSA_TUNING"] = "1"
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rectified_spaattn_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "rectified_spaattn_amd", "librsa_hip_diag.so")
from bench import REGIMES, WORKLOADS, gen_inputs, make_neighbors, make_spec  # noqa: E402
from rectified_spaattn_amd import _core  # noqa: E402


def main():
    L = _lib.lib()
    dev = torch.device("cuda:0")
    wl = WORKLOADS[os.environ.get("RSA_PERF_WORKLOAD", "hunyuan_720p_128f")]      # (cogvideox_768p_81f: the head-dim-64 kernel)
    H = int(os.environ.get("RSA_PERF_H", str(wl["H"])))
    spec = make_spec(wl)
    cent, nbr_kind, p = REGIMES["r2"]
    q, k, v = gen_inputs(wl, H, 0, dev, cent, D=wl.get("D", 128))
    call = _core.StagedCall(q, k, v, spec, wl["top_k"], p, make_neighbors(wl, spec, nbr_kind))
    call.select()
    torch.cuda.synchronize()
    nwg = 8 + H * 2 * 16 + H * ((spec.NBv + 7) // 8 * 8) + 64 + 1024   # (+ the pieces of a split tail)
    dbg = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
    ptr = dbg.data_ptr()
    assert L.rsa_set_tuning(b"k5_w64", 3) == 0
    assert L.rsa_set_tuning(b"dbg_lo", ctypes.c_int(ptr & 0xFFFFFFFF).value) == 0
    assert L.rsa_set_tuning(b"dbg_hi", ctypes.c_int(ptr >> 32).value) == 0
    for gs in (0, 1):
        assert L.rsa_set_tuning(b"k5_gsync", gs) == 0
        for _ in range(3):
            dbg.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); call.attend(); b.record()
            torch.cuda.synchronize()
        d = dbg.view(-1, 8).cpu().numpy()
        d = d[d[:, 4] > 0]
        sparse = d[(d[:, 4] < 200) & (d[:, 4] > 50) & (d[:, 6] > 0)]     # visual query blocks (about 92 kept blocks)
        loop_sub = sparse[:, 1] / (4.0 * sparse[:, 6])
        tail_sub = sparse[:, 2] / (4.0 * (sparse[:, 4] - sparse[:, 6]))
        tot_sub = sparse[:, 5] / (4.0 * sparse[:, 4])
        outside = sparse[:, 5] - sparse[:, 0] - sparse[:, 1] - sparse[:, 2]
        prol, epil = sparse[:, 7] >> 32, sparse[:, 7] & 0xFFFFFFFF
        vmw = (sparse[:, 3] & 0xFFFFFFFF) / (4.0 * sparse[:, 6])
        barw = ((sparse[:, 3] >> 32) & 0xFFFFFF) / (4.0 * sparse[:, 6])
        resc = ((sparse[:, 3] >> 56) & 0xFF)
        q = lambda x, f: sorted(x)[int(len(x) * f)]  # noqa: E731
        print(f"w64 diag build, aligned starts {gs}: {a.elapsed_time(b):.3f} ms | waves {len(sparse)} | kept blocks {sparse[:, 4].mean():.1f}, in the asm loop "
              f"{sparse[:, 6].mean():.1f} | cycles per 32-key sub-step: asm loop {loop_sub.mean():.0f} (p10 {q(loop_sub, .1):.0f} p90 {q(loop_sub, .9):.0f}) "
              f"of which parked on vmcnt {vmw.mean():.0f} (p90 {q(vmw, .9):.0f}), on the barrier {barw.mean():.0f} (p90 {q(barw, .9):.0f}) incl. ~70 per stamp; "
              f"rescales per wave {resc.mean():.2f} | C++-driven tail {tail_sub.mean():.0f} | whole kernel / sub-step {tot_sub.mean():.0f} | prologue + epilogue per wave {outside.mean():.0f} (prologue incl. the wait for the generation {prol.mean():.0f} p50 {q(prol, .5):.0f} p90 {q(prol, .9):.0f} max {prol.max():.0f}, epilogue {epil.mean():.0f})",
              flush=True)
    L.rsa_set_tuning(b"dbg_lo", 0); L.rsa_set_tuning(b"dbg_hi", 0)


if __name__ == "__main__":
    main()
