"""Diagnostic script (uses the oracle, hence kept under tests/): run on the GPU box with python tests/diag/diag_fp8_smooth_k.py."""
import sys, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc
os.environ.setdefault("RSA_TUNING", "1")
from rectified_spaattn_amd import _core, _lib, synth
import test_gpu_fp8 as T
lay = orc.layout_wan(6 * 128, 0)
q, k, v = synth.structured_qkv(515, 1, 2, lay.S, 128, smooth=0.0)
for scale in (0.0, 2.0, 6.0, 12.0):
    bias = np.random.default_rng(5).standard_normal(128).astype(np.float32) * scale
    kb = k + bias[None, None, None, :]
    tq, tk, tv = (torch.from_numpy(x).to("cuda:0", torch.bfloat16) for x in (q, kb, v))
    qf, kf, vf = (t.float().cpu().numpy() for t in (tq, tk, tv))
    ref16 = orc.rectified_attention(qf, kf, vf, lay, 99, 1.5, None)
    call = _core.StagedCall(tq, tk, tv, T._spec(lay), 99, 1.5, None, qkv_fp8=True)
    call.select()
    smooth = call.attend().float().cpu().numpy().reshape(ref16.shape)
    _lib.lib().rsa_set_tuning(b"fp8_smooth_k", 0)
    call.select()
    plain = call.attend().float().cpu().numpy().reshape(ref16.shape)
    _lib.lib().rsa_set_tuning(b"fp8_smooth_k", 1)
    print(f"K bias {scale:4.1f} sigma: mean|dO| vs bf16 oracle: smooth-K {np.abs(smooth-ref16).mean():.3e}  plain {np.abs(plain-ref16).mean():.3e}")
