"""Not collected by pytest: randomized sweep of the dense kernel (fullattn's device path): random Sq / Sk, two-segment
splits, causal, head dim 64 / 128, bf16 / fp16, and the e4m3 form, against the oracle.  python tests/diag/sweep_dense.py <seed> <n>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from rectified_spaattn_amd import _core  # noqa: E402

DEV = "cuda:0"


def main():
    seed, n = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.default_rng(seed)
    bad = 0
    for i in range(n):
        D = int(rng.choice([64, 128]))
        Sq, Sk = int(rng.integers(1, 1500)), int(rng.integers(1, 1500))
        H = int(rng.integers(1, 3))
        f16 = bool(rng.integers(0, 2))
        causal = bool(rng.integers(0, 3) == 0)
        two = bool(rng.integers(0, 3) == 0)
        qs = int(rng.integers(0, Sq + 1)) if two else None
        ks = int(rng.integers(0, Sk + 1)) if two else None
        fp8 = bool(rng.integers(0, 2))
        g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
        dt = torch.float16 if f16 else torch.bfloat16
        q = torch.randn(1, H, Sq, D, generator=g).to(DEV, dt)
        k = torch.randn(1, H, Sk, D, generator=g).to(DEV, dt)
        v = torch.randn(1, H, Sk, D, generator=g).to(DEV, dt)
        out = _core.dense_attention(q, k, v, qs, ks, qkv_fp8=fp8, causal=causal).float().cpu().numpy()   # [1, Sq, H, D]
        msgs = []
        if not np.isfinite(out).all():
            msgs.append("non-finite")
        for h in range(H):
            qf, kf, vf = (t[0, h].float().cpu().numpy() for t in (q, k, v))
            if fp8:
                ref = orc.dense_attention_fp8(qf, kf, vf, qs, ks, causal=causal)
                mx = 8e-2
            else:
                if two:
                    ref = np.zeros((Sq, D))
                    if qs > 0:
                        ref[:qs] = orc.dense_attention(qf[:qs], kf[:ks], vf[:ks], causal=causal) if ks > 0 else 0.0
                    if qs < Sq:
                        ref[qs:] = orc.dense_attention(qf[qs:], kf[ks:], vf[ks:], causal=causal) if ks < Sk else 0.0
                else:
                    ref = orc.dense_attention(qf, kf, vf, causal=causal)
                mx = 2e-3 if f16 else 2e-2
            err = np.abs(out[0, :, h] - ref).max()
            if fp8:   # rows that see a handful of keys reproduce V, whose e4m3 step is 6 % of |v|: judge rows relative to themselves
                rel = np.linalg.norm(out[0, :, h] - ref, axis=-1) / np.maximum(np.linalg.norm(ref, axis=-1), 1e-3)
                if not rel.max() <= 0.2:
                    msgs.append(f"head {h} row-relative max {rel.max():.3e} (abs max {err:.3e})")
            elif not err <= mx:
                msgs.append(f"head {h} max {err:.3e}")
        if msgs:
            bad += 1
            print(f"FAIL case {i}: D={D} Sq={Sq} Sk={Sk} H={H} f16={f16} causal={causal} split=({qs},{ks}) fp8={fp8}: " + "; ".join(msgs), flush=True)
    print(f"dense sweep seed {seed}: {n} cases, {bad} failed")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
