"""Print the fp8 K5's output errors against the fp8-aware oracle and the bf16 oracle for the test cases."""
import sys, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc
from rectified_spaattn_amd import _core, synth
import test_gpu_fp8 as T
for name, mk, H, top_k, p, nbw, dt in T.CASES:
    lay = mk()
    q, k, v = synth.structured_qkv(4242 + len(name), 1, H, lay.S, 128, smooth=0.0)
    nbr = synth.banded_neighbors(lay.NBv, nbw) if nbw >= 0 else None
    tq, tk, tv = (torch.from_numpy(x).to("cuda:0", dt) for x in (q, k, v))
    q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
    tn = torch.from_numpy(nbr) if nbr is not None else None
    o = _core.rectified_attention(tq, tk, tv, T._spec(lay), top_k, p, tn, qkv_fp8=True).float().cpu().numpy()
    o16 = _core.rectified_attention(tq, tk, tv, T._spec(lay), top_k, p, tn).float().cpu().numpy()
    r8 = orc.rectified_attention_fp8(q, k, v, lay, top_k, p, nbr)
    r16 = orc.rectified_attention(q, k, v, lay, top_k, p, nbr)
    e8, e16, eo = np.abs(o - r8), np.abs(o - r16), np.abs(r8 - r16)
    print(f"{name:14s} vs fp8 oracle max {e8.max():.3e} mean {e8.mean():.3e} | vs bf16 oracle max {e16.max():.3e} mean {e16.mean():.3e}"
          f" | fp8 oracle vs bf16 oracle (quantisation alone) max {eo.max():.3e} mean {eo.mean():.3e} | bf16 kernel vs bf16 oracle max {np.abs(o16-r16).max():.3e}"
          f" | rms(out) {np.sqrt((r16**2).mean()):.3f}")
