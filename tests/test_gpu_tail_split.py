"""Tail split of the 64-row K5 (rsa_attn.hip::launch_attn, tuning key k5_tail_split; round 4): when the last generation of 512
workgroups is less than half full, the walks of its query blocks are split over the idle slots (split-KV partials + a combine pass
that also rectifies).  The split blocks' accumulation order changes, so they agree with the unsplit kernel within rounding, not
byte for byte; every other block stays byte-identical; everything agrees with the oracle within the operator's tolerance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _qkv(H, S, seed, dt=torch.bfloat16):
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    nb = (S + 127) // 128
    cent = torch.randn(H, nb, 128, generator=g, device="cuda:0")

    def mk():
        return (cent.repeat_interleave(128, 1)[:, :S] + 0.7 * torch.randn(H, S, 128, generator=g, device="cuda:0")).to(dt).view(1, H, S, 128)
    return mk(), mk(), torch.randn(1, H, S, 128, generator=g, device="cuda:0").to(dt)


def _run(q, k, v, spec, top_k, split):
    from rectified_spaattn_amd import _core, _lib
    L = _lib.lib()
    try:
        assert L.rsa_set_tuning(b"k5_tail_split", split) == 0
        out = _core.rectified_attention(q, k, v, spec, top_k, 0.05, None, shape_xfuse=True)
        torch.cuda.synchronize()
        return out
    finally:
        L.rsa_set_tuning(b"k5_tail_split", 1)


@pytest.mark.parametrize("layout,H,nbv,dt", [("wan", 8, 72, torch.bfloat16), ("wan", 9, 70, torch.float16),
                                             ("hunyuan", 6, 102, torch.bfloat16), ("wan", 4, 140, torch.bfloat16)])
def test_split_tail_against_the_unsplit_kernel_and_the_oracle(layout, H, nbv, dt):
    """wan 8 x 72 = 576 workgroups: one full generation + 64 blocks split 4 ways; 9 x 72 (70 blocks padded to 72) = 648: 136 blocks
    3 ways (with the padding workgroups of the mapping inside the tail); hunyuan 6 x 104 = 624: 112 blocks 4 ways with the text
    pieces behind them; 4 x 144 = 576 again with longer lists."""
    from oracle import oracle as orc
    from rectified_spaattn_amd import _core
    if layout == "wan":
        S = nbv * 128 - 37
        spec, lay = _core.LayoutSpec.wan(S, 2), orc.layout_wan(S, 2)
    else:
        S = (nbv + 2) * 128
        spec, lay = _core.LayoutSpec.hunyuan(S, S - 56), orc.layout_hunyuan(S, S - 56)
    top_k = 12
    q, k, v = _qkv(H, S, 41 + nbv, dt)
    whole = _run(q, k, v, spec, top_k, 0)
    split = _run(q, k, v, spec, top_k, 1)
    again = _run(q, k, v, spec, top_k, 1)
    assert torch.equal(split, again)                                    # deterministic
    assert torch.isfinite(split.float()).all()
    diff = (split.float() - whole.float()).abs()
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    assert diff.max() <= 2 * ulp * max(1.0, float(whole.float().abs().max()))
    # which blocks may differ at all: the sparse blocks behind the last full generation (work index = head * NBp + j, query
    # block (j & 7) * NBp / 8 + (j >> 3))
    NBp = (spec.NBv + 7) // 8 * 8
    n_sparse = H * NBp
    first = (n_sparse // 512) * 512
    assert 0 < n_sparse - first <= 256, "the case does not have a tail to split"
    touched = torch.zeros(H, spec.NB_total, dtype=torch.bool)
    for vv in range(first, n_sparse):
        h, j = divmod(vv, NBp)
        qb = (j & 7) * (NBp // 8) + (j >> 3)
        if qb < spec.NBv:
            touched[h, qb] = True
    blk_diff = diff[0].reshape(-1, 128, H, 128).amax(dim=(1, 3)).t().cpu() if S % 128 == 0 else None
    if blk_diff is not None:
        assert (blk_diff[~touched[:, :blk_diff.shape[1]]] == 0).all(), "a block outside the tail changed"
    assert int(touched.sum()) > 0
    # the oracle, one head that has split blocks
    hd = int(torch.nonzero(touched.any(1))[-1])
    qf, kf, vf = (x[0, hd].float().cpu().numpy() for x in (q, k, v))
    ref = orc.rectified_attention(qf[None, None], kf[None, None], vf[None, None], lay, top_k, 0.05, None)
    got = split[0, :, hd].float().cpu().numpy()
    err = np.abs(got - ref.reshape(got.shape))
    mx, mean = (2e-2, 2e-3) if dt == torch.bfloat16 else (2e-3, 2e-4)
    assert err.max() <= mx and err.mean() <= mean


def test_without_the_partial_buffer_or_with_a_full_last_generation_nothing_is_split():
    """8 x 64 = 512 workgroups exactly, and 8 x 80 = 640 with 128 < 256 in the tail but no tpart (the C entry with buffers that lack
    it): same bytes with the switch on and off."""
    from rectified_spaattn_amd import _core
    q, k, v = _qkv(8, 64 * 128, 51)
    spec = _core.LayoutSpec.wan(64 * 128, 0)
    assert torch.equal(_run(q, k, v, spec, 10, 0), _run(q, k, v, spec, 10, 1))
    # 8 x 80 = 640 workgroups: 128 walks in the tail -- split 3 ways when the buffers carry tpart, whole when they do not
    q, k, v = _qkv(8, 80 * 128, 52)
    spec = _core.LayoutSpec.wan(80 * 128, 0)
    whole = _run(q, k, v, spec, 10, 0)
    assert not torch.equal(whole, _run(q, k, v, spec, 10, 1)), "the case has no tail to split"

    def staged(tpart, nbytes):
        c = _core.StagedCall(q, k, v, spec, 10, 0.05)
        c.cb.tpart, c.cb.tpart_bytes = tpart(c), nbytes(c)
        c.select()
        o = c.attend()
        torch.cuda.synchronize()
        return o
    full_bytes = lambda c: c.bufs["tpart"].numel() * 4            # noqa: E731
    assert torch.equal(staged(lambda c: None, lambda c: 0), whole), "without tpart every walk must stay whole"
    # a partial buffer too small for the 128 x 3 pieces: the split is dropped, not overrun (rsa_buffers.tpart_bytes, 0.5.0)
    assert torch.equal(staged(lambda c: c.bufs["tpart"].data_ptr(), lambda c: 100 * 128 * 130 * 4), whole)
    assert not torch.equal(staged(lambda c: c.bufs["tpart"].data_ptr(), full_bytes), whole)
    # ... and a non-NULL tpart whose capacity is not declared is refused
    from rectified_spaattn_amd import _lib
    with pytest.raises(_lib.RsaError, match="workspace"):
        staged(lambda c: c.bufs["tpart"].data_ptr(), lambda c: 0)


def test_text_split_is_clamped_to_the_declared_capacity():
    """Hunyuan layout, 3 heads, 72 key blocks (4 text pieces by default: 72 / 16): with room for only 2 pieces per text block
    the text rows are split 2 ways instead -- same result within rounding, nothing written past the declared bytes (the region
    behind them stays untouched)."""
    from rectified_spaattn_amd import _core
    H, nbv = 3, 70
    S = (nbv + 2) * 128
    q, k, v = _qkv(H, S, 61)
    spec = _core.LayoutSpec.hunyuan(S, S - 56)
    ref = _core.rectified_attention(q, k, v, spec, 8, 0.05, None, shape_xfuse=True)
    c = _core.StagedCall(q, k, v, spec, 8, 0.05)
    per_piece = 128 * 130 * 4
    cap = H * 2 * 2 * per_piece                   # 2 pieces per text block
    c.bufs["tpart"].fill_(-7.0)
    c.cb.tpart_bytes = cap
    c.select()
    o = c.attend()
    torch.cuda.synchronize()
    assert (c.bufs["tpart"].view(-1)[cap // 4:] == -7.0).all(), "K5 wrote past the declared capacity of tpart"
    assert float((o.float() - ref.float()).abs().max()) <= 2 * 2.0 ** -7 * max(1.0, float(ref.float().abs().max()))


@pytest.mark.parametrize("mode", [True, "pv"], ids=["e4m3", "pv"])
def test_split_tail_in_the_e4m3_kernel(mode):
    """The same split in the e4m3 K5 and in its pv form (head dim 128): split and whole walks agree within the rounding of a different
    summation order, blocks outside the tail byte for byte, and the result is deterministic."""
    from rectified_spaattn_amd import _core, _lib
    H, nbv, top_k = 8, 72, 12
    S = nbv * 128
    q, k, v = _qkv(H, S, 77)
    spec = _core.LayoutSpec.wan(S, 2)
    L = _lib.lib()
    outs = {}
    try:
        for split in (0, 1, 1):
            assert L.rsa_set_tuning(b"k5_tail_split", split) == 0
            o = _core.rectified_attention(q, k, v, spec, top_k, 0.05, None, shape_xfuse=True, qkv_fp8=mode)
            torch.cuda.synchronize()
            outs.setdefault(split, []).append(o)
    finally:
        L.rsa_set_tuning(b"k5_tail_split", 1)
    whole, split = outs[0][0], outs[1][0]
    assert torch.equal(split, outs[1][1])
    diff = (split.float() - whole.float()).abs()
    assert 0 < float(diff.max()) <= 2 * 2.0 ** -7 * max(1.0, float(whole.float().abs().max()))
    NBp = (spec.NBv + 7) // 8 * 8
    first = (H * NBp // 512) * 512
    touched = torch.zeros(H, spec.NB_total, dtype=torch.bool)
    for vv in range(first, H * NBp):
        h, j = divmod(vv, NBp)
        qb = (j & 7) * (NBp // 8) + (j >> 3)
        if qb < spec.NBv:
            touched[h, qb] = True
    blk_diff = diff[0].reshape(-1, 128, H, 128).amax(dim=(1, 3)).t().cpu()
    assert (blk_diff[~touched] == 0).all() and (blk_diff[touched] > 0).any()


@pytest.mark.parametrize("layout,H,nbv,top_k", [("wan", 5, 130, 9), ("wan", 3, 200, 20), ("wan", 11, 56, 6), ("flux", 5, 116, 10),
                                                ("hunyuan", 3, 190, 12), ("wan", 7, 100, 50), ("wan", 2, 300, 8)])
def test_split_plans_of_every_shape(layout, H, nbv, top_k):
    """Different tail sizes and piece counts (2, 3, 4 pieces; tails with and without text pieces behind them; lists shorter than
    the piece count would like; a case the planner must leave alone): split == whole within rounding, finite, deterministic."""
    from rectified_spaattn_amd import _core
    if layout == "wan":
        S = nbv * 128 - 5
        spec = _core.LayoutSpec.wan(S, 1)
    elif layout == "flux":
        S = (nbv + 4) * 128
        spec = _core.LayoutSpec.flux(S, 512)
    else:
        S = (nbv + 2) * 128
        spec = _core.LayoutSpec.hunyuan(S, S - 100)
    q, k, v = _qkv(H, S, 100 + H + nbv)
    whole = _run(q, k, v, spec, top_k, 0)
    split = _run(q, k, v, spec, top_k, 1)
    assert torch.isfinite(split.float()).all()
    assert torch.equal(split, _run(q, k, v, spec, top_k, 1))
    diff = (split.float() - whole.float()).abs()
    assert diff.max() <= 2 * 2.0 ** -7 * max(1.0, float(whole.float().abs().max()))
