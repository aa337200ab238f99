"""K3b (union lists of query-block pairs) and K5's paired 256-row workgroups.

The paired form must be invisible in the results: the same bytes as the 128-row form (every query block still walks
exactly its own kept blocks in ascending order), whatever mixture of paired / unpaired blocks the overlap test picks."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(q, k, v, spec, top_k, p, nbr, pair):
    from rectified_spaattn_amd import _core, _lib
    assert _lib.lib().rsa_set_tuning(b"k5_pair", pair) == 0
    try:
        out, bufs = _core.rectified_attention(q, k, v, spec, top_k, p, nbr, return_parts=True)
        torch.cuda.synchronize()
    finally:
        _lib.lib().rsa_set_tuning(b"k5_pair", 0)
    return out, bufs


@pytest.mark.parametrize("case", [
    # (variant, B, H, S, D, top_k, p, neighbour band, smooth)
    ("wan", 1, 2, 2048, 128, 4, 0.6, 2, 0.8),      # smooth centroids: heavy overlap, most pairs ok
    ("wan", 2, 2, 1450, 128, 3, 0.3, 1, 0.0),      # padded tail block, odd number of blocks
    ("hunyuan", 1, 2, 3328, 128, 5, 0.3, 1, 0.5),  # text tail kept by every row
    ("wan", 1, 3, 1100, 64, 2, 0.5, -1, 0.0),      # head_dim 64, no neighbours: little overlap, mixed pairs
    ("cogvideo", 1, 2, 994, 64, 2, 0.3, 1, 0.9),
])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_paired_form_is_bit_identical(case, dt):
    from rectified_spaattn_amd import _core, synth
    variant, B, H, S, D, top_k, p, band, smooth = case
    q, k, v = (torch.from_numpy(x).to(DEV, dt) for x in synth.structured_qkv(77, B, H, S, D, smooth=smooth))
    if variant == "wan":
        spec = _core.LayoutSpec.wan(S, 1)
    elif variant == "hunyuan":
        spec = _core.LayoutSpec.hunyuan(S, S - 56)
    else:
        spec = _core.LayoutSpec.cogvideo(S, 226)
    nbr = torch.from_numpy(synth.banded_neighbors(spec.NBv, band)) if band >= 0 else None
    o1, b1 = _run(q, k, v, spec, top_k, p, nbr, 1)
    o0, b0 = _run(q, k, v, spec, top_k, p, nbr, 0)
    assert torch.equal(o1, o0), float((o1.float() - o0.float()).abs().max())
    # K3b against a numpy restatement from the bitmask
    kept = _core.unpack_bitmask(b1["bitmask"], spec.NB_total).cpu().numpy()          # [BH, NBv, NB]
    pcols = b1["pcols"].cpu().numpy().view(np.uint16)
    pcounts, pair_ok = b1["pcounts"].cpu().numpy(), b1["pair_ok"].cpu().numpy()
    NP = (spec.NBv + 1) // 2
    n_ok = 0
    for bh in range(kept.shape[0]):
        for pi in range(NP):
            a = kept[bh, 2 * pi]
            b = kept[bh, 2 * pi + 1] if 2 * pi + 1 < spec.NBv else np.zeros_like(a)
            u = np.nonzero(a | b)[0]
            want = u | (a[u].astype(np.int64) << 14) | (b[u].astype(np.int64) << 15)
            assert pcounts[bh, pi] == u.size
            assert np.array_equal(pcols[bh, pi, :u.size].astype(np.int64), want)
            ok = 2 * pi + 1 < spec.NBv and 10 * int((a & b).sum()) >= 3 * max(int(a.sum()), int(b.sum()))
            assert pair_ok[bh, pi] == int(ok)
            n_ok += int(ok)
    print(f"{case}: {n_ok} of {kept.shape[0] * NP} pairs served by paired workgroups")


def test_paired_form_runs_in_the_golden_style_case_and_matches_the_oracle():
    """A case where every pair is ok: the output still meets the oracle tolerance (the 128-row form is not involved)."""
    from rectified_spaattn_amd import _core, synth
    S, D, top_k, p = 16 * 128, 128, 6, 0.7
    q, k, v = synth.structured_qkv(5, 1, 2, S, D, smooth=0.95)
    lay = orc.layout_wan(S, 2)
    nbr = synth.banded_neighbors(lay.NBv, 3)
    tq, tk, tv = (torch.from_numpy(x).to(DEV, torch.bfloat16) for x in (q, k, v))
    out, bufs = _run(tq, tk, tv, _core.LayoutSpec.wan(S, 2), top_k, p, torch.from_numpy(nbr), 1)
    assert int(bufs["pair_ok"].sum()) >= bufs["pair_ok"].numel() // 2
    ref = orc.rectified_attention(q, k, v, lay, top_k, p, nbr)
    err = np.abs(out.float().cpu().numpy() - ref)
    assert err.max() <= 2e-2 and err.mean() <= 2e-3
