#!/usr/bin/env python3
"""Where K5's sub-steps spend their cycles: loads the DIAGNOSTICS build (make -C rectified_spaattn_amd/csrc diag ->
librsa_hip_diag.so: s_memtime at four points of every sub-step, differences summed per wave in scalar registers) and
prints, per K5 form (tuning key k5_form), the mean cycles per sub-step in [vmcnt wait + barrier | DMA issue + rare
branches | pipelined block], for the sparse R2 call and a dense call.  The diagnostic build's fences forbid overlaps the
real kernel has: read the SHARES, not the totals (cdna_hip_programming.md section 7)."""
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
    H = int(os.environ.get("RSA_PERF_H", "24"))
    wl = WORKLOADS["hunyuan_720p_128f"]
    spec = make_spec(wl)
    cent, nbr_kind, p = REGIMES["r2"]
    q, k, v = gen_inputs(wl, H, 0, dev, cent)
    call = _core.StagedCall(q, k, v, spec, wl["top_k"], p, make_neighbors(wl, spec, nbr_kind))
    call.select()
    torch.cuda.synchronize()
    nwg = 8 + H * 2 * 16 + H * ((spec.NBv + 7) // 8 * 8) + 64
    dbg = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
    ptr = dbg.data_ptr()
    for blk in [int(x) for x in os.environ.get("RSA_DIAG_FORMS", "0,1,2").split(",")]:
        assert L.rsa_set_tuning(b"k5_form", blk) == 0
        assert L.rsa_set_tuning(b"dbg_lo", ctypes.c_int(ptr & 0xFFFFFFFF).value) == 0
        assert L.rsa_set_tuning(b"dbg_hi", ctypes.c_int(ptr >> 32).value) == 0
        for _ in range(2):
            dbg.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); call.attend(); b.record()
            torch.cuda.synchronize()
        d = dbg.view(-1, 8).cpu().numpy()
        d = d[d[:, 4] > 0]
        sparse = d[d[:, 4] < 400]       # visual query blocks (92 kept blocks = 184 tiles); text blocks walk their split
        steps = 2.0 * sparse[:, 4]
        w, h, bl, tot = (sparse[:, i] / steps for i in range(4))
        print(f"k5_form={blk}: {a.elapsed_time(b):.3f} ms (diag build) | waves {len(sparse)} | per sub-step: wait+barrier {w.mean():.0f} "
              f"(p10 {sorted(w)[len(w)//10]:.0f} p90 {sorted(w)[len(w)*9//10]:.0f}) | dma+head {h.mean():.0f} | block {bl.mean():.0f} "
              f"(p10 {sorted(bl)[len(bl)//10]:.0f} p90 {sorted(bl)[len(bl)*9//10]:.0f}) | whole kernel / sub-step {tot.mean():.0f} | "
              f"outside the loop per workgroup {(sparse[:, 3] - sparse[:, 0] - sparse[:, 1] - sparse[:, 2]).mean():.0f}", flush=True)
        for wv in range(4):
            sel = sparse[wv::4] if len(sparse) % 4 == 0 else sparse
            st = 2.0 * sel[:, 4]
            print(f"    wave {wv}: wait {(sel[:,0]/st).mean():.0f} head {(sel[:,1]/st).mean():.0f} block {(sel[:,2]/st).mean():.0f}")
    L.rsa_set_tuning(b"dbg_lo", 0); L.rsa_set_tuning(b"dbg_hi", 0)


if __name__ == "__main__":
    main()
