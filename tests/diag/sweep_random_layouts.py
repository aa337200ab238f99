"""Not collected by pytest: a wider one-off sweep of test_gpu_random_layouts' generator (other seeds, up to 40 visual blocks,
both operand paths) for hunting rare failures on the GPU box:  python tests/diag/sweep_random_layouts.py <seed> <count>
Prints one line per failing case and a summary; exit code 1 if anything failed."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from rectified_spaattn_amd import _core, synth  # noqa: E402

DEV = "cuda:0"


def cases(seed, n, nbv_max=40):
    rng = np.random.default_rng(seed)
    for i in range(n):
        variant = ["wan", "hunyuan", "flux", "cogvideo"][int(rng.integers(0, 4))]
        D = int(rng.choice([16, 32, 64, 128]))   # 16 / 32: the zero-padded path
        H = int(rng.integers(1, 3))
        nbv = int(rng.integers(1, nbv_max + 1))
        if variant == "wan":
            S = int(nbv * 128 - rng.integers(0, 127))
            lay = orc.layout_wan(S, int(rng.integers(0, nbv + 3)))
        elif variant == "hunyuan":
            S = nbv * 128 + 256
            lay = orc.layout_hunyuan(S, nbv * 128 + int(rng.integers(1, 257)))
        elif variant == "flux":
            tl = int(rng.choice([128, 256, 512]))
            S = nbv * 128 + tl
            lay = orc.layout_flux(S, tl)
        else:
            tl = int(rng.integers(1, 256))
            pad = int(rng.integers(0, 128))
            tl = tl if (tl + pad) % 128 == 0 else 256 - pad
            S = nbv * 128 + tl
            lay = orc.layout_cogvideo(S, tl)
        top_k = int(rng.integers(0, min(lay.L + 3, 12)))
        p = float(rng.choice([0.0, 0.05, 0.1, 0.3, 0.6, 0.95, 1.5]))
        nbw = int(rng.integers(-1, 3))
        yield (i, variant, D, H, lay, top_k, p, nbw, int(rng.integers(0, 1 << 30)), bool(rng.integers(0, 2)))


def run(case):
    i, variant, D, H, lay, top_k, p, nbw, dseed, f16 = case
    q, k, v = synth.structured_qkv(dseed, 1, H, lay.S, D, smooth=0.5 if i % 3 == 0 else 0.0)
    nbr = synth.banded_neighbors(lay.NBv, nbw) if nbw >= 0 else None
    dt = torch.float16 if f16 else torch.bfloat16
    tq, tk, tv = (torch.from_numpy(x).to(DEV, dt) for x in (q, k, v))
    q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
    spec = _core.LayoutSpec(lay.S, lay.NB_total, lay.NBv, lay.n_txt, lay.kv_valid, lay.pool_valid,
                            lay.text_end_block, lay.ffb, lay.q_text_valid, lay.kv_text_valid)
    tn = torch.from_numpy(nbr) if nbr is not None else None
    out, bufs = _core.rectified_attention(tq, tk, tv, spec, top_k, p, tn, return_parts=True)
    ref, parts = orc.rectified_attention(q, k, v, lay, top_k, p, nbr, want_parts=True)
    msgs = []
    for bh in range(H):
        sel = parts[bh]
        kept = orc.unpack_bits(bufs["bitmask"][bh].cpu().numpy().view(np.uint32), lay.NB_total)
        for name, a, b in (("mask", kept, sel["kept"]), ("unrel", bufs["unrel"][bh].cpu().numpy(), sel["unrel"]),
                           ("probs", bufs["probs"][bh].cpu().numpy(), sel["probs"]),
                           ("R", bufs["R"][bh].cpu().numpy(), sel["R"])):
            if not np.array_equal(a, b):
                msgs.append(f"{name} head {bh}")
    mx, mean = (2e-3, 2e-4) if f16 else (2e-2, 2e-3)
    err = np.abs(out.float().cpu().numpy() - ref)
    if not (err.max() <= mx and err.mean() <= mean and np.isfinite(err).all()):
        msgs.append(f"O max {err.max():.3e} mean {err.mean():.3e}")
    if D in (64, 128):
        o8, p8 = _core.rectified_attention(tq, tk, tv, spec, top_k, p, tn, return_parts=True, qkv_fp8=True)
        ref8, _, ops = orc.rectified_attention_fp8(q, k, v, lay, top_k, p, nbr, want_parts=True)
        for n_ in ("q8", "k8", "v8t"):
            if not np.array_equal(p8[n_].cpu().numpy(), ops[n_]):
                msgs.append(f"fp8 image {n_}")
        if not np.array_equal(p8["exps"].cpu().numpy().astype(np.uint32), ops["exps"]):
            msgs.append("fp8 exponents")
        e8 = np.abs(o8.float().cpu().numpy() - ref8)
        # (max: a row carried by one or two keys reproduces V, whose e4m3 step is 6 % of |v|: 8e-2 here, mean unchanged)
        if not (e8.max() <= 8e-2 and e8.mean() <= 6e-3 and np.isfinite(e8).all()):
            msgs.append(f"fp8 O max {e8.max():.3e} mean {e8.mean():.3e}")
    return msgs


if __name__ == "__main__":
    seed, count = int(sys.argv[1]), int(sys.argv[2])
    bad = 0
    for c in cases(seed, count):
        m = run(c)
        if m:
            bad += 1
            print(f"FAIL case {c[0]} {c[1]} D={c[2]} H={c[3]} S={c[4].S} NBv={c[4].NBv} top_k={c[5]} p={c[6]} nbw={c[7]} "
                  f"seed={c[8]} f16={c[9]}: " + "; ".join(m), flush=True)
    print(f"sweep seed {seed}: {count} cases, {bad} failed")
    sys.exit(1 if bad else 0)
