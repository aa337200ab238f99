"""Extreme but valid layouts through the operator, against the oracle: 1 and 256 valid text tokens (HunyuanVideo), one visual block,
first_frame_blocks beyond the row, neighbour matrices all True / all False, a text tail with no visual block.
Run on the GPU box:  timeout 300 python tests/diag/diag_degenerate_layouts.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as orc  # noqa: E402
from rectified_spaattn_amd import _core, synth  # noqa: E402

DEV = "cuda:0"


def spec_of(lay):
    return _core.LayoutSpec(lay.S, lay.NB_total, lay.NBv, lay.n_txt, lay.kv_valid, lay.pool_valid, lay.text_end_block, lay.ffb,
                            lay.q_text_valid, lay.kv_text_valid)


def case(name, lay, top_k, p, nbr=None, H=2, D=128, dt=torch.bfloat16):
    q, k, v = synth.structured_qkv(1234 + lay.S, 1, H, lay.S, D, smooth=0.0)
    tq, tk, tv = (torch.from_numpy(x).to(DEV, dt) for x in (q, k, v))
    q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
    try:
        out, parts = _core.rectified_attention(tq, tk, tv, spec_of(lay), top_k, p, torch.from_numpy(nbr) if nbr is not None else None,
                                               return_parts=True)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        print(f"{name}: device raised {type(e).__name__}: {str(e)[:120]}")
        return
    try:
        ref, sel = orc.rectified_attention(q, k, v, lay, top_k, p, nbr, want_parts=True)
    except Exception as e:  # noqa: BLE001
        print(f"{name}: device finite={bool(torch.isfinite(out).all())}; oracle raised {type(e).__name__}: {str(e)[:120]}")
        return
    kept = np.stack([s["kept"] for s in sel])
    got = _core.unpack_bitmask(parts["bitmask"], lay.NB_total).cpu().numpy().reshape(kept.shape)
    err = np.abs(out.float().cpu().numpy() - ref)
    print(f"{name}: mask equal {np.array_equal(got, kept)}, kept fraction {kept.mean():.3f}, max|d| {err.max():.2e} mean {err.mean():.2e} "
          f"finite={bool(torch.isfinite(out).all())}")


if __name__ == "__main__":
    S = 6 * 128 + 256
    case("hunyuan 1 valid text token", orc.layout_hunyuan(S, 6 * 128 + 1), 2, 0.3)
    case("hunyuan 256 valid text tokens", orc.layout_hunyuan(S, S), 2, 0.3)
    case("hunyuan 128 valid text tokens", orc.layout_hunyuan(S, 6 * 128 + 128), 2, 0.3)
    case("hunyuan 129 valid text tokens", orc.layout_hunyuan(S, 6 * 128 + 129), 2, 0.3)
    case("hunyuan one visual block", orc.layout_hunyuan(128 + 256, 128 + 77), 1, 0.3)
    case("flux text 512, one visual block", orc.layout_flux(128 + 512, 512), 1, 0.2)
    case("flux text 128", orc.layout_flux(5 * 128 + 128, 128), 2, 0.2)
    case("cogvideo text 226, S % 128 == 0", orc.layout_cogvideo(7 * 128, 226), 2, 0.3)
    case("cogvideo text 226, one visual block", orc.layout_cogvideo(128 + 226, 226), 1, 0.3)
    case("wan ffb beyond the row", orc.layout_wan(5 * 128 + 3, 99), 1, 0.1)
    case("wan ffb = NB", orc.layout_wan(5 * 128, 5), 1, 0.1)
    n = 5
    case("wan neighbours all True", orc.layout_wan(n * 128, 0), 1, 0.0, np.ones((n, n), np.bool_))
    case("wan neighbours all False", orc.layout_wan(n * 128, 0), 1, 0.0, np.zeros((n, n), np.bool_))
    case("wan fp16 head dim 64", orc.layout_wan(3 * 128 + 50, 1), 1, 0.2, None, 2, 64, torch.float16)
    print("done")
