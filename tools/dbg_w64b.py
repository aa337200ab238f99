#!/usr/bin/env python3
import os, sys
os.environ["RSA_TUNING"] = "1"
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_op_case, case_inputs
from rectified_spaattn_amd import _core, _lib
from test_gpu_parity import _spec
L = _lib.lib()
name = "wan_640"
meta, gold = load_op_case(name)
q, k, v, lay, nbr = case_inputs(meta)
dt = torch.float16
for vmode in ("real", "ones", "real", "real"):
    vv = v if vmode == "real" else np.ones_like(v)
    tq, tk, tv = (torch.from_numpy(x).to("cuda:0", dt) for x in (q, k, vv))
    outs = []
    for w in (0, 3):
        assert L.rsa_set_tuning(b"k5_w64", w) == 0
        out, bufs = _core.rectified_attention(tq, tk, tv, _spec(lay), meta["top_k"], meta["p"],
                                              torch.from_numpy(nbr) if nbr is not None else None, return_parts=True)
        torch.cuda.synchronize()
        outs.append(out.float().cpu().numpy().reshape(1, 640, 2, 128).copy())
    cols = bufs["cols"].cpu().numpy(); cnt = bufs["counts"].cpu().numpy()
    print(vmode, "lists head0:", [cols[0, i, :cnt[0, i]].tolist() for i in range(5)], "R", bufs["R"][0].cpu().numpy())
    a, b = outs
    print("   per q-block max|diff| head0/1:", [float(np.abs(a[0, i*128:(i+1)*128] - b[0, i*128:(i+1)*128]).max()) for i in range(5)])
L.rsa_set_tuning(b"k5_w64", 0)
