"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C-ABI library
(rectified_spaattn_amd/librsa_hip.so); the oracle (oracle/) and the golden vectors are the checkers.

Tolerances: block mask / GAPR mask / kept lists bit-exact; fp32 statistics bit-exact against
the contract (oracle/rsa_oracle.c); attention output vs fp64 oracle on the same 2-byte inputs:
bf16 max|d| <= 2e-2, mean|d| <= 2e-3; fp16 max|d| <= 2e-3."""
import numpy as np
import pytest
import torch

from conftest import B2_CASES, BIG_CASES, OP_CASES, case_inputs, load_op_case, reference_rows
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

TOL = {torch.bfloat16: (2e-2, 2e-3), torch.float16: (2e-3, 2e-4)}


def _spec(lay):
    from rectified_spaattn_amd import _core
    return _core.LayoutSpec(lay.S, lay.NB_total, lay.NBv, lay.n_txt, lay.kv_valid, lay.pool_valid,
                            lay.text_end_block, lay.ffb, lay.q_text_valid, lay.kv_text_valid)


def _run(q, k, v, lay, top_k, p, nbr, dt):
    from rectified_spaattn_amd import _core
    dev = torch.device("cuda:0")
    tq, tk, tv = (torch.from_numpy(x).to(dev, dt) for x in (q, k, v))
    seen = tuple(x.float().cpu().numpy() for x in (tq, tk, tv))
    out, bufs = _core.rectified_attention(tq, tk, tv, _spec(lay), top_k, p,
                                          torch.from_numpy(nbr) if nbr is not None else None, return_parts=True)
    torch.cuda.synchronize()
    return out.float().cpu().numpy(), {n: t.cpu().numpy() for n, t in bufs.items()}, seen


@pytest.mark.parametrize("name", OP_CASES + B2_CASES)
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_operator_vs_oracle_and_golden(name, dt):
    meta, gold = load_op_case(name)
    q, k, v, lay, nbr = case_inputs(meta)
    out, g, (q, k, v) = _run(q, k, v, lay, meta["top_k"], meta["p"], nbr, dt)
    ref, parts = orc.rectified_attention(q, k, v, lay, meta["top_k"], meta["p"], nbr, want_parts=True)
    B, H = meta["B"], meta["H"]
    for bh in range(B * H):
        sel, st = parts[bh], parts[bh]["stats"]
        for nm, want in (("qbar", st.qbar), ("aq", st.aq), ("kbar", st.kbar), ("ak", st.ak), ("vbar", st.vbar)):
            assert np.array_equal(g[nm][bh], want), f"{name} {nm} not bit-exact"
        assert np.array_equal(g["scores"][bh][:, : lay.NBv], sel["s_vis"])
        if lay.n_txt:
            assert np.array_equal(g["scores"][bh][:, lay.NBv:], sel["s_txt"])
        assert np.array_equal(g["unrel"][bh], sel["unrel"])
        assert np.array_equal(g["probs"][bh], sel["probs"])
        kept = orc.unpack_bits(g["bitmask"][bh].view(np.uint32), lay.NB_total)
        assert np.array_equal(kept, sel["kept"]), f"{name}: block mask differs from the oracle"
        if dt == torch.bfloat16:  # goldens were produced from the bf16-representable inputs
            assert np.array_equal(kept, gold["one_hot"][bh // H, bh % H]), f"{name}: block mask differs from reference"
            assert np.array_equal(g["unrel"][bh], gold["nogapr"][bh // H, bh % H])
            np.testing.assert_allclose(g["probs"][bh], gold["probs"][bh // H, bh % H], rtol=2e-5, atol=1e-6)
        cnt = sel["kept"].sum(-1)
        assert np.array_equal(g["counts"][bh], cnt)
        for i in range(lay.NBv):
            assert np.array_equal(g["cols"][bh][i, : cnt[i]], np.nonzero(sel["kept"][i])[0])
        assert np.array_equal(g["R"][bh], sel["R"])
        assert np.array_equal(g["w"][bh], sel["w"])
        np.testing.assert_allclose(g["comp"][bh], sel["comp"], atol=1e-5)
    mx, mean = TOL[dt]
    err = np.abs(out - ref)
    assert err.max() <= mx and err.mean() <= mean, f"{name}: max {err.max():.3e} mean {err.mean():.3e}"
    if dt == torch.bfloat16:
        e2 = np.abs(out - gold["out"])[reference_rows(meta, lay)]
        assert e2.max() <= mx and e2.mean() <= mean


@pytest.mark.parametrize("name", BIG_CASES)
def test_long_row_case_vs_reference_vectors(name):
    """Rows of 260 columns (K3's sorted-head path, 20 kept + neighbours per row): every mask bit, GAPR bit and
    probability against the reference's own run, O against its output (stored as fp16)."""
    from rectified_spaattn_amd import _core
    meta, gold = load_op_case(name)
    q, k, v, lay, nbr = case_inputs(meta)
    out, g, _ = _run(q, k, v, lay, meta["top_k"], meta["p"], nbr, torch.bfloat16)
    kept = orc.unpack_bits(g["bitmask"][0].view(np.uint32), lay.NB_total)
    assert np.array_equal(kept, gold["one_hot"][0, 0]), f"{name}: block mask differs from the reference"
    assert np.array_equal(g["unrel"][0], gold["nogapr"][0, 0])
    np.testing.assert_allclose(g["probs"][0], gold["probs"][0, 0], rtol=2e-5, atol=1e-6)
    err = np.abs(out - gold["out"].astype(np.float32))
    assert err.max() <= 2e-2 and err.mean() <= 2e-3, f"{name}: max {err.max():.3e} mean {err.mean():.3e}"


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(1, 2, 640, 128), (2, 3, 1000, 128), (1, 2, 777, 64), (1, 1, 100, 128)])
def test_dense_kernel(shape, dt):
    from rectified_spaattn_amd import _core, synth
    B, H, S, D = shape
    q, k, v = synth.structured_qkv(5, B, H, S, D)
    dev = torch.device("cuda:0")
    tq, tk, tv = (torch.from_numpy(x).to(dev, dt) for x in (q, k, v))
    q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
    out = _core.dense_attention(tq, tk, tv).float().cpu().numpy()  # [B,S,H,D]
    mx, mean = TOL[dt]
    for b in range(B):
        for h in range(H):
            ref = orc.dense_attention(q[b, h], k[b, h], v[b, h])
            err = np.abs(out[b, :, h] - ref)
            assert err.max() <= mx and err.mean() <= mean
    # two-segment varlen semantics (attn.py:107-120): rows < qs see kv < ks ; rows >= qs see kv >= ks
    qs, ks = S - 37, S - 50
    out2 = _core.dense_attention(tq, tk, tv, q_split=qs, kv_split=ks).float().cpu().numpy()
    for b in range(B):
        for h in range(H):
            r1 = orc.dense_attention(q[b, h][:qs], k[b, h][:ks], v[b, h][:ks])
            r2 = orc.dense_attention(q[b, h][qs:], k[b, h][ks:], v[b, h][ks:])
            assert np.abs(out2[b, :qs, h] - r1).max() <= mx
            assert np.abs(out2[b, qs:, h] - r2).max() <= mx


def test_strided_views_and_output_layout():
    """q/k/v given as [B,H,S,D] views of [B,S,H,D] memory (what the processors produce) need no copy."""
    from rectified_spaattn_amd import _core, synth
    B, H, S, D = 1, 2, 1024, 128
    q, k, v = synth.structured_qkv(9, B, H, S, D)
    dev = torch.device("cuda:0")
    mk = lambda x: torch.from_numpy(x).to(dev, torch.bfloat16).permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
    tq, tk, tv = mk(q), mk(k), mk(v)
    assert not tq.is_contiguous()
    lay = orc.layout_wan(S, 1)
    out = _core.rectified_attention(tq, tk, tv, _spec(lay), 3, 0.3, None)
    ref = orc.rectified_attention(q, k, v, lay, 3, 0.3, None)
    assert out.shape == (B, S, H * D)
    assert np.abs(out.float().cpu().numpy() - ref).max() <= 2e-2


def test_keep_all_equals_dense_at_scale():
    """Size-independent property at a larger shape: top_k = NB  =>  R == 1, comp == 0, output == dense kernel."""
    from rectified_spaattn_amd import _core, synth
    B, H, S, D = 1, 2, 128 * 40 + 13, 128
    q, k, v = synth.structured_qkv(21, B, H, S, D)
    dev = torch.device("cuda:0")
    tq, tk, tv = (torch.from_numpy(x).to(dev, torch.bfloat16) for x in (q, k, v))
    spec = _core.LayoutSpec.wan(S, 0)
    out, bufs = _core.rectified_attention(tq, tk, tv, spec, spec.NBv, 0.3, None, return_parts=True)
    dense = _core.dense_attention(tq, tk, tv).reshape(B, S, H * D)
    assert bool((bufs["counts"] == spec.NB_total).all())
    assert float((bufs["R"] - 1).abs().max()) < 1e-5
    assert float(bufs["comp"].abs().max()) == 0.0
    assert float((out.float() - dense.float()).abs().max()) <= 1e-2


def test_online_softmax_rescale_branch():
    """Spiked keys late in the sequence force the running max to jump (rescale path) in a late tile."""
    from rectified_spaattn_amd import _core
    torch.manual_seed(0)
    B, H, S, D = 1, 1, 1024, 128
    q = torch.randn(B, H, S, D)
    k = torch.randn(B, H, S, D) * 0.1
    v = torch.randn(B, H, S, D)
    k[0, 0, 900] = q[0, 0, 5] * 4.0   # row 5's max jumps at key 900
    k[0, 0, 70] = q[0, 0, 300] * 3.0
    dev = torch.device("cuda:0")
    tq, tk, tv = (x.to(dev, torch.bfloat16) for x in (q, k, v))
    out = _core.dense_attention(tq, tk, tv).float().cpu().numpy()[0, :, 0]
    ref = orc.dense_attention(*(x.float().cpu().numpy()[0, 0] for x in (tq, tk, tv)))
    assert np.abs(out - ref).max() <= 2e-2


def test_library_rejects_bad_arguments():
    from rectified_spaattn_amd import _core, _lib
    dev = torch.device("cuda:0")
    q = torch.zeros(1, 1, 256, 96, dtype=torch.bfloat16, device=dev)
    with pytest.raises(AssertionError):
        _core.dense_attention(q, q, q)
    q32 = torch.zeros(1, 1, 256, 128, dtype=torch.float32, device=dev)
    with pytest.raises(AssertionError):
        _core.dense_attention(q32, q32, q32)
    qc = torch.zeros(1, 1, 256, 128, dtype=torch.bfloat16)
    with pytest.raises(_lib.RsaError):
        _core.dense_attention(qc, qc, qc)


@pytest.mark.parametrize("Sq,Sk,causal", [(1, 1, False), (1, 129, False), (127, 1, False), (129, 257, False), (1, 1, True), (5, 300, True)])
def test_dense_degenerate_shapes(Sq, Sk, causal):
    """One query row, one key, a ragged block on either side: the right numbers (fullattn, attn.py:101-120)."""
    from rectified_spaattn_amd import _core
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Sq * 1000 + Sk)
    q = torch.randn(1, 2, Sq, 128, generator=g).to(dev, torch.bfloat16)
    k = torch.randn(1, 2, Sk, 128, generator=g).to(dev, torch.bfloat16)
    v = torch.randn(1, 2, Sk, 128, generator=g).to(dev, torch.bfloat16)
    out = _core.dense_attention(q, k, v, causal=causal).float().cpu().numpy()      # [B, Sq, H, D]
    for h in range(2):
        ref = orc.dense_attention(*(x[0, h].float().cpu().numpy() for x in (q, k, v)), causal=causal)
        assert np.abs(out[0, :, h] - ref).max() <= 2e-2, (h, np.abs(out[0, :, h] - ref).max())


def test_dense_empty_operands_raise():
    """No query rows or no keys: an exception from the library's argument check, never a launch."""
    from rectified_spaattn_amd import _core, _lib
    dev = torch.device("cuda:0")
    mk = lambda s: torch.zeros(1, 2, s, 128, dtype=torch.bfloat16, device=dev)   # noqa: E731
    for Sq, Sk in ((0, 128), (128, 0), (0, 0)):
        with pytest.raises(_lib.RsaError):
            _core.dense_attention(mk(Sq), mk(Sk), mk(Sk))


@pytest.mark.parametrize("S,top_k,p", [(1, 1, 0.5), (100, 0, 0.0), (128, 1, 0.0), (129, 1, 0.0), (300, 0, 0.0), (300, 99, 1.0),
                                       (300, 1, -1.0), (300, 1, 2.0)])
def test_operator_degenerate_shapes_and_parameters(S, top_k, p):
    """A one-token sequence, a single ragged block, top_k = 0 / larger than the row, p outside [0, 1]: the oracle's answer
    (n = max(#{cumsum <= p} + 1, top_k) clamped to the row: rectified_wan21_attn.py:230-235)."""
    from rectified_spaattn_amd import synth
    lay = orc.layout_wan(S, 0)
    q, k, v = synth.structured_qkv(77 + S, 1, 2, S, 128, smooth=0.0)
    out, bufs, (q, k, v) = _run(q, k, v, lay, top_k, p, None, torch.bfloat16)
    ref, parts = orc.rectified_attention(q, k, v, lay, top_k, p, None, want_parts=True)
    assert np.isfinite(out).all()
    err = np.abs(out - ref)
    assert err.max() <= 2e-2 and err.mean() <= 2e-3, (err.max(), err.mean())
    kept = np.stack([sel["kept"] for sel in parts])
    from rectified_spaattn_amd import _core
    got = _core.unpack_bitmask(torch.from_numpy(bufs["bitmask"]), lay.NB_total).numpy()
    assert np.array_equal(got.reshape(kept.shape), kept), "block mask"


_S_H = 6 * 128 + 256
EXTREME_LAYOUTS = [
    ("hunyuan_1_text_token", lambda: orc.layout_hunyuan(_S_H, 6 * 128 + 1), 2, 0.3, None),
    ("hunyuan_256_text_tokens", lambda: orc.layout_hunyuan(_S_H, _S_H), 2, 0.3, None),
    ("hunyuan_128_text_tokens", lambda: orc.layout_hunyuan(_S_H, 6 * 128 + 128), 2, 0.3, None),
    ("hunyuan_129_text_tokens", lambda: orc.layout_hunyuan(_S_H, 6 * 128 + 129), 2, 0.3, None),
    ("hunyuan_one_visual_block", lambda: orc.layout_hunyuan(128 + 256, 128 + 77), 1, 0.3, None),
    ("flux_one_visual_block", lambda: orc.layout_flux(128 + 512, 512), 1, 0.2, None),
    ("flux_text_128", lambda: orc.layout_flux(5 * 128 + 128, 128), 2, 0.2, None),
    ("cogvideo_one_visual_block", lambda: orc.layout_cogvideo(128 + 226, 226), 1, 0.3, None),
    ("wan_ffb_beyond_the_row", lambda: orc.layout_wan(5 * 128 + 3, 99), 1, 0.1, None),
    ("wan_ffb_equals_nb", lambda: orc.layout_wan(5 * 128, 5), 1, 0.1, None),
    ("wan_neighbours_all_true", lambda: orc.layout_wan(5 * 128, 0), 1, 0.0, True),
    ("wan_neighbours_all_false", lambda: orc.layout_wan(5 * 128, 0), 1, 0.0, False),
]


@pytest.mark.parametrize("case", EXTREME_LAYOUTS, ids=lambda c: c[0])
def test_operator_extreme_layouts(case):
    """The corners of each layout's geometry (hunyuan :313-332, flux :307-320, cogvideo :306-322, wan21 :297-313, :265-274): one
    and 256 valid text tokens, a text tail that ends on / one past a block edge, a single visual block, first_frame_blocks at and
    beyond the row, neighbour matrices that keep everything / nothing: mask bit-exact, O within the bf16 bound."""
    from rectified_spaattn_amd import _core, synth
    name, mk, top_k, p, nb = case
    lay = mk()
    nbr = None if nb is None else np.full((lay.NBv, lay.NBv), nb, np.bool_)
    q, k, v = synth.structured_qkv(1234 + lay.S, 1, 2, lay.S, 128, smooth=0.0)
    out, bufs, (q, k, v) = _run(q, k, v, lay, top_k, p, nbr, torch.bfloat16)
    ref, parts = orc.rectified_attention(q, k, v, lay, top_k, p, nbr, want_parts=True)
    kept = np.stack([sel["kept"] for sel in parts])
    got = _core.unpack_bitmask(torch.from_numpy(bufs["bitmask"]), lay.NB_total).numpy().reshape(kept.shape)
    assert np.array_equal(got, kept), "block mask"
    err = np.abs(out - ref)
    assert np.isfinite(out).all() and err.max() <= 2e-2 and err.mean() <= 2e-3, (err.max(), err.mean())
