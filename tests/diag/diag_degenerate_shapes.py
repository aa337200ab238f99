"""Degenerate shapes through the public entry points: each must return the right numbers or raise -- never hang or crash.
Run on the GPU box under `timeout`:  timeout 300 python tests/diag/diag_degenerate_shapes.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as orc  # noqa: E402
from rectified_spaattn_amd import _core, _lib  # noqa: E402

DEV = "cuda:0"


def dense_case(Sq, Sk, D=128, dt=torch.bfloat16, **kw):
    g = torch.Generator().manual_seed(Sq * 1000 + Sk)
    q = torch.randn(1, 2, Sq, D, generator=g).to(DEV, dt)
    k = torch.randn(1, 2, Sk, D, generator=g).to(DEV, dt)
    v = torch.randn(1, 2, Sk, D, generator=g).to(DEV, dt)
    try:
        o = _core.dense_attention(q, k, v, **kw)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        return f"raised {type(e).__name__}: {str(e)[:100]}"
    if Sq == 0 or Sk == 0:
        return f"returned shape {tuple(o.shape)} finite={bool(torch.isfinite(o).all())}"
    ref = np.stack([orc.dense_attention(q[0, h].float().cpu().numpy(), k[0, h].float().cpu().numpy(), v[0, h].float().cpu().numpy(),
                                        causal=bool(kw.get("causal", False)))
                    for h in range(2)])
    got = o.float().cpu().numpy().reshape(1, Sq, 2, D)[0].transpose(1, 0, 2) if o.dim() == 3 else o[0].float().cpu().numpy().transpose(1, 0, 2) if o.shape[1] == Sq else o[0].float().cpu().numpy()
    return f"max|d| {np.abs(got - ref).max():.2e}"


def sparse_case(S, top_k, p, H=2, D=128, nbr=None, ffb=0):
    from rectified_spaattn_amd import synth
    lay = orc.layout_wan(S, ffb)
    q, k, v = synth.structured_qkv(77 + S, 1, H, S, D, smooth=0.0)
    tq, tk, tv = (torch.from_numpy(x).to(DEV, torch.bfloat16) for x in (q, k, v))
    spec = _core.LayoutSpec(lay.S, lay.NB_total, lay.NBv, lay.n_txt, lay.kv_valid, lay.pool_valid, lay.text_end_block, lay.ffb,
                            lay.q_text_valid, lay.kv_text_valid)
    try:
        out = _core.rectified_attention(tq, tk, tv, spec, top_k, p, nbr)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        return f"raised {type(e).__name__}: {str(e)[:100]}"
    q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
    try:
        ref = orc.rectified_attention(q, k, v, lay, top_k, p, None)
    except Exception as e:  # noqa: BLE001
        return f"device returned finite={bool(torch.isfinite(out).all())}; oracle raised {type(e).__name__}: {str(e)[:80]}"
    return f"max|d| {np.abs(out.float().cpu().numpy() - ref).max():.2e} finite={bool(torch.isfinite(out).all())}"


if __name__ == "__main__":
    for Sq, Sk in ((1, 1), (1, 129), (127, 1), (129, 257), (0, 128), (128, 0), (0, 0)):
        print(f"dense Sq={Sq} Sk={Sk}:", dense_case(Sq, Sk))
    for Sq, Sk in ((1, 1), (5, 300)):
        print(f"dense causal Sq={Sq} Sk={Sk}:", dense_case(Sq, Sk, causal=True))
    for S, tk, p in ((1, 1, 0.5), (100, 0, 0.0), (128, 1, 0.0), (129, 1, 0.0), (300, 0, 0.0), (300, 99, 1.0), (300, 1, -1.0), (300, 1, 2.0)):
        print(f"sparse wan S={S} top_k={tk} p={p}:", sparse_case(S, tk, p))
    print("done")
