#!/usr/bin/env python3
"""Study (CPU, oracle only): how much of the fp8 operand path's distance to the bf16 oracle is due to the GRANULARITY
of the e4m3 scales?  Quantises Q, K (minus its mean), V with (a) one scale per head (what rsa_fp8.hip did until round 3), (b) one
power-of-two scale per 128-token block, (c) one power-of-two scale per token row, dequantises, and runs the exact (fp64)
rectified attention on those values -- so the numbers isolate the operand rounding (P is NOT rounded here).
    python tests/diag/diag_fp8_scale_granularity.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as orc  # noqa: E402
from rectified_spaattn_amd import synth  # noqa: E402


def pow2_scale(amax):
    """smallest power of two s with amax / s <= 448"""
    a = np.maximum(np.asarray(amax, np.float64), 1e-30)
    return np.exp2(np.ceil(np.log2(a / 448.0))).astype(np.float32)


def quant(x, scale):
    return orc.dequantize_e4m3(orc.quantize_e4m3((x / scale).astype(np.float32))) * scale


def variants(q, k, v):
    """q, k, v [S, D] fp32 (one head) -> dict name -> (qd, kd, vd)"""
    mu = k.mean(0, keepdims=True).astype(np.float32)
    kc = k - mu
    S, D = q.shape
    nb = (S + 127) // 128
    pad = nb * 128 - S

    def blocks(x):
        xp = np.pad(x, ((0, pad), (0, 0)))
        return xp.reshape(nb, 128, D)

    out = {}
    out["per-head"] = tuple(quant(x, np.float32(np.abs(x).max() / 448.0)) for x in (q, kc, v))
    res = []
    for x in (q, kc, v):
        xb = blocks(x)
        sc = pow2_scale(np.abs(xb).max(axis=(1, 2)))[:, None, None]
        res.append(quant(xb, sc).reshape(-1, D)[:S])
    out["per-128-block pow2"] = tuple(res)
    res = []
    for x in (q, kc, v):
        sc = pow2_scale(np.abs(x).max(axis=1))[:, None]
        res.append(quant(x, sc))
    out["per-row pow2"] = tuple(res)
    return out


def main():
    rows = []
    for name, lay, top_k, p, nbw, seed, spike in (
            ("hunyuan 6+2 blocks", orc.layout_hunyuan(6 * 128 + 256, 6 * 128 + 200), 2, 0.4, 1, 4249, 0.0),
            ("wan 12 blocks", orc.layout_wan(12 * 128, 1), 3, 0.3, 1, 99, 0.0),
            ("wan 12 blocks, outlier channels", orc.layout_wan(12 * 128, 1), 3, 0.3, 1, 99, 6.0)):
        q, k, v = (x[0, 0] for x in synth.structured_qkv(seed, 1, 1, lay.S, 128))
        if spike:   # a few channels / tokens with large magnitude, as real activations have
            q, k, v = q.copy(), k.copy(), v.copy()
            k[:, 5] *= spike
            q[:, 5] *= 0.5
            v[::97] *= spike
            q, k, v = (orc.round_bf16(x) for x in (q, k, v))
        nbr = synth.banded_neighbors(lay.NBv, nbw)
        ref = orc.rectified_attention(q[None, None], k[None, None], v[None, None], lay, top_k, p, nbr)
        kk, vv = k.copy(), v.copy()
        kk[lay.pool_valid:] = 0
        vv[lay.pool_valid:] = 0
        sel = orc.select_head(q, kk, vv, lay, top_k, p, nbr)
        for vname, (qd, kd, vd) in variants(q, kk, vv).items():
            o = orc.sparse_attention_head(qd, kd, vd, lay, sel["kept"], sel["rows"])
            o = o * sel["R"][:, None, None].astype(np.float64) + sel["comp"][:, None, :].astype(np.float64)
            o = o.reshape(-1, 128)[: min(lay.S, lay.NBv * 128)]
            e = np.abs(o - ref[0, : o.shape[0]])
            rows.append((name, vname, e.max(), e.mean(), float(np.sqrt((ref ** 2).mean()))))
    print(f"{'case':34s} {'scales':22s} {'max|dO|':>9s} {'mean|dO|':>9s} {'|O|rms':>7s}")
    for r in rows:
        print(f"{r[0]:34s} {r[1]:22s} {r[2]:9.3e} {r[3]:9.3e} {r[4]:7.3f}")


if __name__ == "__main__":
    main()
