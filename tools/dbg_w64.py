#!/usr/bin/env python3
"""Debug: one golden operator case through the product K5 and the 64-row K5, per query block / row-half differences."""
import os, sys
os.environ["RSA_TUNING"] = "1"
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_op_case, case_inputs
from rectified_spaattn_amd import _core, _lib
from test_gpu_parity import _spec

L = _lib.lib()
for name in sys.argv[1:] or ["wan_640"]:
    meta, gold = load_op_case(name)
    q, k, v, lay, nbr = case_inputs(meta)
    for dt in (torch.bfloat16, torch.float16):
        tq, tk, tv = (torch.from_numpy(x).to("cuda:0", dt) for x in (q, k, v))
        outs = []
        for w in (0, 1, 3):
            assert L.rsa_set_tuning(b"k5_w64", w) == 0
            out, bufs = _core.rectified_attention(tq, tk, tv, _spec(lay), meta["top_k"], meta["p"],
                                                  torch.from_numpy(nbr) if nbr is not None else None, return_parts=True)
            torch.cuda.synchronize()
            outs.append(out.float().cpu().numpy().copy())
        cnt = bufs["counts"].cpu().numpy()
        d = np.abs(outs[0] - outs[1])      # [B, S, H, D]
        print(name, dt, "max diff (asm loop)", d.max(), " (C++-driven only)", np.abs(outs[0] - outs[2]).max(), "counts", cnt.tolist())
        H = meta["H"]
        d = d.reshape(d.shape[0], d.shape[1], H, -1)
        S = d.shape[1]
        for h in range(H):
            per = [d[0, i * 32:(i + 1) * 32, h].max() for i in range((S + 31) // 32)]
            print("  head", h, "per 32-row group:", " ".join(f"{x:.1e}" for x in per))
L.rsa_set_tuning(b"k5_w64", 0)
