#!/usr/bin/env python3
import ctypes, os, sys
os.environ["RSA_TUNING"] = "1"
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rectified_spaattn_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "rectified_spaattn_amd", "librsa_hip_diag.so")
from conftest import load_op_case, case_inputs
from rectified_spaattn_amd import _core
from test_gpu_parity import _spec
L = _lib.lib()
meta, gold = load_op_case("wan_640")
q, k, v, lay, nbr = case_inputs(meta)
dt = torch.float16
tq, tk, tv = (torch.from_numpy(x).to("cuda:0", dt) for x in (q, k, v))
dbg = torch.zeros(1 << 20, dtype=torch.int64, device="cuda:0")
ptr = dbg.data_ptr()
for w in (3, 1):
    assert L.rsa_set_tuning(b"k5_w64", w) == 0
    assert L.rsa_set_tuning(b"dbg_lo", ctypes.c_int(ptr & 0xFFFFFFFF).value) == 0
    assert L.rsa_set_tuning(b"dbg_hi", ctypes.c_int(ptr >> 32).value) == 0
    dbg.zero_()
    out, bufs = _core.rectified_attention(tq, tk, tv, _spec(lay), meta["top_k"], meta["p"], torch.from_numpy(nbr), return_parts=True)
    torch.cuda.synchronize()
    f = dbg[(1 << 16):].view(torch.float32).cpu().numpy().reshape(-1, 128, 8)
    st = dbg[: 16 * 4 * 8].cpu().numpy().reshape(16, 4, 8)
    for work in range(16):
        if st[work, 0, 4] == 0: continue
        print(f"mode {w} work {work}: items {st[work,0,4]} in-loop {st[work,0,6]} | lane0: m_ref {f[work,0,0]:.3f} {f[work,0,1]:.3f} thr {f[work,0,2]} {f[work,0,3]} "
              f"l {f[work,0,4]:.3e} {f[work,0,5]:.3e} mxA {f[work,0,6]:.3f} mxB {f[work,0,7]:.3f} | lane 70: m_ref {f[work,70,0]:.3f} l {f[work,70,4]:.3e} | out row0 {out.float().reshape(1,640,2,128)[0, 0 if work%8==0 else 128*(work%8), work//8, :2].tolist()}")
L.rsa_set_tuning(b"dbg_lo", 0); L.rsa_set_tuning(b"dbg_hi", 0); L.rsa_set_tuning(b"k5_w64", 0)
