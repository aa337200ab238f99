"""Randomised small layouts (GPU): every discrete / fp32 output of the selection kernels must equal the oracle bit
for bit, the output must be within tolerance -- including the awkward corners (one visual block, one text token,
top_k 0 or > row length, first-frame square larger than the block count, S not a multiple of 128, tiny S)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cases():
    rng = np.random.default_rng(20251212)
    out = []
    for i in range(28):
        variant = ["wan", "hunyuan", "flux", "cogvideo"][i % 4]
        D = int(rng.choice([64, 128]))
        H = int(rng.integers(1, 4))
        nbv = int(rng.integers(1, 13))
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
            tl = tl if (tl + pad) % 128 == 0 else 256 - pad  # keep the text tail block-aligned like the scripts
            S = nbv * 128 + tl
            lay = orc.layout_cogvideo(S, tl)
        top_k = int(rng.integers(0, lay.L + 3))
        p = float(rng.choice([0.0, 0.1, 0.3, 0.6, 0.95, 1.5]))
        nbw = int(rng.integers(-1, 3))
        out.append((i, variant, D, H, lay, top_k, p, nbw))
    return out


@pytest.mark.parametrize("case", _cases(), ids=lambda c: f"{c[0]}-{c[1]}-D{c[2]}")
def test_random_layout(case):
    from rectified_spaattn_amd import _core, synth
    i, variant, D, H, lay, top_k, p, nbw = case
    q, k, v = synth.structured_qkv(1000 + i, 1, H, lay.S, D, smooth=0.5 if i % 3 == 0 else 0.0)
    nbr = synth.banded_neighbors(lay.NBv, nbw) if nbw >= 0 else None
    dt = torch.bfloat16 if i % 2 == 0 else torch.float16
    tq, tk, tv = (torch.from_numpy(x).to(DEV, dt) for x in (q, k, v))
    q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
    spec = _core.LayoutSpec(lay.S, lay.NB_total, lay.NBv, lay.n_txt, lay.kv_valid, lay.pool_valid,
                            lay.text_end_block, lay.ffb, lay.q_text_valid, lay.kv_text_valid)
    out, bufs = _core.rectified_attention(tq, tk, tv, spec, top_k, p, torch.from_numpy(nbr) if nbr is not None else None,
                                          return_parts=True)
    ref, parts = orc.rectified_attention(q, k, v, lay, top_k, p, nbr, want_parts=True)
    for bh in range(H):
        sel = parts[bh]
        kept = orc.unpack_bits(bufs["bitmask"][bh].cpu().numpy().view(np.uint32), lay.NB_total)
        assert np.array_equal(kept, sel["kept"]), f"mask (case {i})"
        assert np.array_equal(bufs["unrel"][bh].cpu().numpy(), sel["unrel"])
        assert np.array_equal(bufs["probs"][bh].cpu().numpy(), sel["probs"])
        assert np.array_equal(bufs["R"][bh].cpu().numpy(), sel["R"])
        assert np.array_equal(bufs["counts"][bh].cpu().numpy(), sel["kept"].sum(-1))
    mx, mean = (2e-2, 2e-3) if dt == torch.bfloat16 else (2e-3, 2e-4)
    err = np.abs(out.float().cpu().numpy() - ref)
    assert err.max() <= mx and err.mean() <= mean, f"case {i}: max {err.max():.3e} mean {err.mean():.3e}"


@pytest.mark.parametrize("shift", [60, 68, 72])
def test_tiny_magnitudes_keep_the_contract(shift):
    """K2 and K4 run their fp32 chains on the matrix pipe (v_mfma_f32_32x32x2_f32).  With Q and K scaled by 2^-shift the
    pooled products fall into the subnormal range (2^-136 ... 2^-144 and below): the GAPR comparison |s| > |e_q| + |e_k|
    is then decided on subnormal fp32 values, which the MFMA must neither flush nor round differently from the oracle's
    fmaf chain.  Everything discrete must still equal the oracle bit for bit."""
    from rectified_spaattn_amd import _core, synth
    D, H = 128, 2
    lay = orc.layout_hunyuan(6 * 128 + 256, 6 * 128 + 77)
    q, k, v = synth.structured_qkv(4242 + shift, 1, H, lay.S, D)
    sc = np.float32(2.0 ** -shift)
    tq, tk, tv = (torch.from_numpy(x).to(DEV, torch.bfloat16) for x in (q * sc, k * sc, v))
    q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
    assert np.abs(q).max() > 0 and np.abs(q).max() < 2.0 ** (-shift + 4)      # bf16 keeps the scaled values (no flush)
    spec = _core.LayoutSpec(lay.S, lay.NB_total, lay.NBv, lay.n_txt, lay.kv_valid, lay.pool_valid,
                            lay.text_end_block, lay.ffb, lay.q_text_valid, lay.kv_text_valid)
    out, bufs = _core.rectified_attention(tq, tk, tv, spec, 3, 0.3, None, return_parts=True)
    ref, parts = orc.rectified_attention(q, k, v, lay, 3, 0.3, None, want_parts=True)
    sub = 0
    for bh in range(H):
        sel = parts[bh]
        s_gpu = bufs["scores"][bh].cpu().numpy()
        sub += int(((np.abs(s_gpu) > 0) & (np.abs(s_gpu) < np.float32(2.0 ** -126))).sum())
        kept = orc.unpack_bits(bufs["bitmask"][bh].cpu().numpy().view(np.uint32), lay.NB_total)
        assert np.array_equal(bufs["unrel"][bh].cpu().numpy(), sel["unrel"])
        assert np.array_equal(bufs["probs"][bh].cpu().numpy(), sel["probs"])
        assert np.array_equal(kept, sel["kept"])
        assert np.array_equal(bufs["R"][bh].cpu().numpy(), sel["R"])
    if shift >= 68:
        assert sub > 0, "the case is meant to produce subnormal pooled scores"
    err = np.abs(out.float().cpu().numpy() - ref)
    assert err.max() <= 2e-2 and err.mean() <= 2e-3


@pytest.mark.parametrize("tl", [640, 768, 1024])
def test_many_text_tokens_take_the_tail_path_of_k3(tl):
    """K3 keeps 64 (KPL + 8) score columns of a row in registers; layouts with more individually scored text tokens than
    that (none of the reference's pipelines: Flux has 512) run the remaining columns through the LDS loop.  Same contract."""
    from rectified_spaattn_amd import _core, synth
    D, H, nbv = 128, 2, 3
    S = nbv * 128 + tl
    lay = orc.layout_flux(S, tl)
    q, k, v = synth.structured_qkv(900 + tl, 1, H, lay.S, D)
    tq, tk, tv = (torch.from_numpy(x).to(DEV, torch.bfloat16) for x in (q, k, v))
    q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
    spec = _core.LayoutSpec(lay.S, lay.NB_total, lay.NBv, lay.n_txt, lay.kv_valid, lay.pool_valid,
                            lay.text_end_block, lay.ffb, lay.q_text_valid, lay.kv_text_valid)
    assert lay.NBv + lay.n_txt > 64 * (1 + 8)     # KPL = 1 for three visual blocks: columns past 576 exist
    out, bufs = _core.rectified_attention(tq, tk, tv, spec, 2, 0.3, None, return_parts=True)
    ref, parts = orc.rectified_attention(q, k, v, lay, 2, 0.3, None, want_parts=True)
    for bh in range(H):
        sel = parts[bh]
        kept = orc.unpack_bits(bufs["bitmask"][bh].cpu().numpy().view(np.uint32), lay.NB_total)
        assert np.array_equal(kept, sel["kept"])
        assert np.array_equal(bufs["probs"][bh].cpu().numpy(), sel["probs"])
        assert np.array_equal(bufs["R"][bh].cpu().numpy(), sel["R"])
    err = np.abs(out.float().cpu().numpy() - ref)
    assert err.max() <= 2e-2 and err.mean() <= 2e-3
