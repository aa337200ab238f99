"""The extreme layouts of tests/test_gpu_parity.py::EXTREME_LAYOUTS through the OTHER forms of the operator: head dim 64 (the 64-row
kernel's head-dim-64 instance), fp16, the one-call entry point, and the two fp8 forms against their own oracles.
Run on the GPU box:  timeout 600 python tests/diag/diag_degenerate_layouts_forms.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc  # noqa: E402
from rectified_spaattn_amd import _core, synth  # noqa: E402
from test_gpu_parity import EXTREME_LAYOUTS, _spec  # noqa: E402

DEV = "cuda:0"
worst = {}


def note(kind, name, val, bound):
    flag = "" if val <= bound else "   <-- ABOVE BOUND"
    worst[kind] = max(worst.get(kind, 0.0), val / bound)
    return f"{val:.2e}{flag}"


for name, mk, top_k, p, nb in EXTREME_LAYOUTS:
    lay = mk()
    nbr = None if nb is None else np.full((lay.NBv, lay.NBv), nb, np.bool_)
    tn = torch.from_numpy(nbr) if nbr is not None else None
    for D, dt in ((64, torch.bfloat16), (64, torch.float16), (128, torch.float16)):
        q, k, v = synth.structured_qkv(4321 + lay.S, 1, 2, lay.S, D, smooth=0.0)
        tq, tk, tv = (torch.from_numpy(x).to(DEV, dt) for x in (q, k, v))
        q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
        ref, parts = orc.rectified_attention(q, k, v, lay, top_k, p, nbr, want_parts=True)
        kept = np.stack([s["kept"] for s in parts])
        out, bufs = _core.rectified_attention(tq, tk, tv, _spec(lay), top_k, p, tn, return_parts=True)
        got = _core.unpack_bitmask(bufs["bitmask"], lay.NB_total).cpu().numpy().reshape(kept.shape)
        e = np.abs(out.float().cpu().numpy() - ref)
        o1, _ = _core.rectified_attention_onecall(tq, tk, tv, _spec(lay), top_k, p, tn)
        same = torch.equal(o1.reshape(out.shape), out)
        bound = 2e-2 if dt == torch.bfloat16 else 2e-3
        print(f"{name} D={D} {str(dt)[6:]}: mask {np.array_equal(got, kept)} max|d| {note('2byte', name, e.max(), bound)} onecall==staged {same}")
        if dt != torch.bfloat16:
            continue
        for mode, qk in ((True, "e4m3"), ("pv", "2byte")):
            o8 = _core.rectified_attention(tq, tk, tv, _spec(lay), top_k, p, tn, qkv_fp8=mode)
            r8c = orc.rectified_attention_fp8(q, k, v, lay, top_k, p, nbr, p_form="code", qk=qk)
            r8 = orc.rectified_attention_fp8(q, k, v, lay, top_k, p, nbr, qk=qk)
            ec, ee = np.abs(o8.float().cpu().numpy() - r8c), np.abs(o8.float().cpu().numpy() - r8)
            print(f"    fp8 {mode}: vs code-map oracle {note('fp8 code', name, ec.max(), 2e-2 if mode is True else 4e-2)}, vs exact-P oracle "
                  f"{note('fp8 exact', name, ee.max(), 4e-2)} finite {bool(torch.isfinite(o8).all())}")
print("worst ratio to bound:", {k_: round(v_, 3) for k_, v_ in worst.items()})
