"""K3's sorted-head ("prefix") path against its full-sort path on whole full-size calls: every row of the bitmask, kept
lists, counts, R and compensation weights must be identical (the oracle pins sampled rows in test_gpu_fullsize.py; this
test covers all 21 600 rows per head configuration, including inputs that force the fall-back: plateaus of equal
probabilities and thresholds the head cannot reach)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _select(q, k, v, spec, top_k, p, nbr, prefix):
    from rectified_spaattn_amd import _core, _lib
    assert _lib.lib().rsa_set_tuning(b"k3_prefix", prefix) == 0
    try:
        call = _core.StagedCall(q, k, v, spec, top_k, p, nbr)
        call.select()
        torch.cuda.synchronize()
    finally:
        _lib.lib().rsa_set_tuning(b"k3_prefix", 1)
    cnt = call.bufs["counts"]
    cols = call.bufs["cols"]
    valid = torch.arange(cols.shape[-1], device=cols.device)[None, None, :] < cnt[..., None]
    return dict(bitmask=call.bufs["bitmask"].clone(), counts=cnt.clone(), cols=torch.where(valid, cols, -1),
                R=call.bufs["R"].clone(), w=call.bufs["w"].clone(), probs=call.bufs["probs"].clone())


@pytest.mark.parametrize("cfg", [
    # (layout, H, S, top_k, p, data)
    ("hunyuan", 2, 115456, 90, 0.05, "structured"),
    ("hunyuan", 2, 115456, 180, 0.3, "structured"),
    ("hunyuan", 1, 115456, 90, 0.9, "iid"),          # flat rows: the sum passes 0.9 far beyond the head -> fall-back
    ("hunyuan", 1, 115456, 90, 0.3, "zeros"),        # every probability equal: no threshold isolates a head
    ("wan", 2, 75600, 147, 0.3, "structured"),
    ("flux", 2, 66048, 51, 0.3, "iid"),
    ("wan", 2, 40000, 250, 0.2, "structured"),       # top_k close to the head capacity
    ("wan", 2, 40000, 300, 0.2, "structured"),       # top_k beyond it: prefix path not applicable
])
def test_prefix_path_equals_full_sort(cfg):
    from bench import gen_qkv
    from rectified_spaattn_amd import _core
    layout, H, S, top_k, p, data = cfg
    D = 128
    dev = torch.device(DEV)
    if data == "structured":
        q, k, v = gen_qkv(H, 0, S, S, D, dev, seed=5)
    elif data == "iid":
        g = torch.Generator(device=dev).manual_seed(3)
        q, k, v = (torch.randn(1, H, S, D, generator=g, device=dev).to(torch.bfloat16) for _ in range(3))
    else:
        q = torch.zeros(1, H, S, D, device=dev, dtype=torch.bfloat16)
        k = torch.ones_like(q)
        v = torch.ones_like(q)
    if layout == "hunyuan":
        spec = _core.LayoutSpec.hunyuan(S, S - 56)
    elif layout == "flux":
        spec = _core.LayoutSpec.flux(S, 512)
    else:
        spec = _core.LayoutSpec.wan(S, 6)
    a = _select(q, k, v, spec, top_k, p, None, 1)
    b = _select(q, k, v, spec, top_k, p, None, 0)
    for name in a:
        assert torch.equal(a[name], b[name]), f"{cfg}: {name} differs between the two K3 paths"
    assert int(a["counts"].min()) >= min(top_k, spec.L)


@pytest.mark.parametrize("fp8", [False, True])
def test_text_rows_split_kv_matches_single_workgroup_form(fp8):
    """K5 splits the key range of the dense text query blocks over up to 16 workgroups (+ a combine kernel); the visual
    rows must not change by a bit, the text rows only by the rounding of a different summation order.  The e4m3 kernel has
    its own partial writer (scaled by s_v, offset by P_OFFSET) in front of the shared combine kernel: same check."""
    from bench import gen_qkv
    from rectified_spaattn_amd import _core, _lib
    S, H, D = 115456, 2, 128
    q, k, v = gen_qkv(H, 0, S, S, D, torch.device(DEV), seed=9)
    spec = _core.LayoutSpec.hunyuan(S, S - 56)
    outs = []
    for flag in (1, 0):
        assert _lib.lib().rsa_set_tuning(b"k5_tsplit", flag) == 0
        try:
            outs.append(_core.rectified_attention(q, k, v, spec, 90, 0.05, None, qkv_fp8=fp8).clone())
            torch.cuda.synchronize()
        finally:
            _lib.lib().rsa_set_tuning(b"k5_tsplit", 1)
    a, b = outs
    nv = spec.NBv * 128
    assert torch.equal(a[:, :nv], b[:, :nv])
    d = (a[:, nv:].float() - b[:, nv:].float()).abs()
    assert float(d.max()) <= (8e-3 if fp8 else 4e-3), float(d.max())
    assert float(a[:, nv + spec.q_text_valid:].abs().max()) == 0.0 and float(a[:, nv:nv + spec.q_text_valid].abs().max()) > 0


def _long_rows_case(nb, text, D=64, top_k=20, p=0.02, rows=None):
    from rectified_spaattn_amd import _core
    from oracle import oracle as orc
    S = (nb + (2 if text else 0)) * 128
    g = torch.Generator(device=DEV).manual_seed(5 + nb)
    cent = torch.randn(S // 128, D, generator=g, device=DEV) * 1.5

    def mk():
        return (cent.repeat_interleave(128, 0) + 0.5 * torch.randn(S, D, generator=g, device=DEV)).to(torch.bfloat16).view(1, 1, S, D)
    q, k = mk(), mk()
    v = torch.randn(1, 1, S, D, generator=g, device=DEV).to(torch.bfloat16)
    spec = _core.LayoutSpec.hunyuan(S, S - 56) if text else _core.LayoutSpec.wan(S, 3)
    lay = orc.layout_hunyuan(S, S - 56) if text else orc.layout_wan(S, 3)
    out, parts = _core.rectified_attention(q, k, v, spec, top_k, p, None, return_parts=True)
    assert torch.isfinite(out.float()).all()
    qf, kf, vf = (x[0, 0].float().cpu().numpy() for x in (q, k, v))
    if lay.pool_valid < lay.S:
        kf[lay.pool_valid:] = 0
        vf[lay.pool_valid:] = 0
    rows = rows or [0, nb // 2 - 1, nb - 1]
    sel = orc.select_head(qf, kf, vf, lay, top_k, p, None, rows=rows)
    kept = _core.unpack_bitmask(parts["bitmask"], lay.NB_total).cpu().numpy()
    for a_, i in enumerate(rows):
        assert np.array_equal(kept[0, i], sel["kept"][a_].astype(bool)), i
        n = int(parts["counts"][0, i])
        assert n == int(sel["kept"][a_].sum())
        assert np.array_equal(parts["cols"][0, i, :n].cpu().numpy(), np.nonzero(sel["kept"][a_])[0])
        assert np.array_equal(parts["probs"][0, i].cpu().numpy(), sel["probs"][a_])
        assert parts["R"][0, i].item() == sel["R"][a_]
        assert np.array_equal(parts["w"][0, i].cpu().numpy(), sel["w"][a_])
    # whole-result properties: lists and bitmask agree on every row, every row keeps at least top_k entries
    cnt = parts["counts"][0].cpu().numpy()
    assert np.array_equal(kept[0].sum(-1), cnt) and (cnt >= top_k).all()
    # the sampled query blocks of O against the oracle
    ref = orc.sparse_attention_head(qf, kf, vf, lay, sel["kept"], rows)
    ref = ref * sel["R"][:, None, None] + sel["comp"][:, None, :]
    o = out.view(1, lay.S, 1, D)
    for a_, i in enumerate(rows):
        err = np.abs(o[0, i * 128:(i + 1) * 128, 0].float().cpu().numpy() - ref[a_])
        assert err.max() <= 2e-2 and err.mean() <= 2e-3


def test_longest_rows_of_the_one_wave_kernel():
    """K3's one-wave-per-row kernel keeps a row's probabilities, text exponentials and keep bytes in 16 KB of LDS per wave:
    2 816 visual blocks (360k tokens) -- mask, lists, probabilities, R, w and O of sampled rows against the oracle."""
    _long_rows_case(2816, text=False)


@pytest.mark.parametrize("nb,text", [(3000, False), (4100, False), (3200, True)])
def test_rows_beyond_it_take_the_workgroup_per_row_kernel(nb, text):
    """Round 5: rows the one-wave kernel cannot hold (refused with 'unsupported' in rounds 2-4; the reference has no limit,
    rectified_hunyuan_attn.py:226-262) run select_mask_long_kernel: 3 000 blocks (a 4 096-key sort), 4 100 blocks = 525k tokens
    (an 8 192-key sort), and a text-carrying layout (IPAR, the collapsed text entry) -- bit for bit against the oracle on
    sampled rows, O within the operator's tolerance."""
    _long_rows_case(nb, text)


def test_refusal_beyond_8192_key_blocks():
    """K5 holds a walk's kept list as u16 entries in 16 KB of LDS: 8 192 key blocks (1 M tokens) is the path's limit; beyond it
    the library answers 'unsupported' (AssertionError, as the reference's own shape asserts) instead of computing wrongly."""
    from rectified_spaattn_amd import _core
    D = 64
    S2 = 8200 * 128
    z = torch.zeros(1, 1, S2, D, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(AssertionError):
        _core.rectified_attention(z, z, z, _core.LayoutSpec.wan(S2, 0), 20, 0.02, None)


@pytest.mark.parametrize("cfg", [("hunyuan", 2, 37 * 128 + 256, 9, 0.3, "structured"), ("wan", 3, 33 * 128 - 19, 5, 0.05, "iid"),
                                 ("flux", 2, 20 * 128 + 512, 4, 0.9, "iid"), ("hunyuan", 1, 40 * 128 + 256, 6, 0.3, "zeros"),
                                 ("hunyuan", 2, 115456, 90, 0.05, "structured")])
def test_workgroup_per_row_kernel_equals_the_one_wave_kernel(cfg):
    """The long-row kernel forced onto shapes the one-wave kernel serves (tuning key k3_long): every output of the selection --
    bitmask, lists, counts, R, w, probabilities -- identical on every row, including plateaus of equal probabilities (zeros),
    thresholds the sum only passes late (p = 0.9), neighbours, text entries and the headline row length."""
    from bench import gen_qkv
    from rectified_spaattn_amd import _core, _lib, synth
    layout, H, S, top_k, p, data = cfg
    D = 128
    dev = torch.device(DEV)
    if data == "structured":
        q, k, v = gen_qkv(H, 0, S, S, D, dev, seed=5)
    elif data == "iid":
        g = torch.Generator(device=dev).manual_seed(3)
        q, k, v = (torch.randn(1, H, S, D, generator=g, device=dev).to(torch.bfloat16) for _ in range(3))
    else:
        q = torch.zeros(1, H, S, D, device=dev, dtype=torch.bfloat16)
        k = torch.ones_like(q)
        v = torch.ones_like(q)
    spec = {"hunyuan": lambda: _core.LayoutSpec.hunyuan(S, S - 56), "flux": lambda: _core.LayoutSpec.flux(S, 512),
            "wan": lambda: _core.LayoutSpec.wan(S, 2)}[layout]()
    nbr = torch.from_numpy(synth.banded_neighbors(spec.NBv, 1)) if S < 100000 else None
    L = _lib.lib()
    res = []
    for flag in (0, 1):
        assert L.rsa_set_tuning(b"k3_long", flag) == 0
        try:
            res.append(_select(q, k, v, spec, top_k, p, nbr, 1))
        finally:
            L.rsa_set_tuning(b"k3_long", 0)
    for name in res[0]:
        assert torch.equal(res[0][name], res[1][name]), f"{cfg}: {name} differs between the one-wave and the workgroup-per-row kernel"


@pytest.mark.parametrize("name", ["hunyuan_1280", "wan_pad_1450", "flux_1536"])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_k5_32row_kernel_still_serves_head_dim_128(name, dt):
    """Head dim 128 runs the 64-rows-per-wave K5 (rsa_attn_kernel64.hip) since round 4; the 32-row kernel (the product at head
    dim 64, reachable at 128 through the tuning key k5_w64 = 0 for A/B) must keep giving the same answer: both against the
    oracle, and against each other within one output ulp (the row sums are added in a different order)."""
    from conftest import case_inputs, load_op_case
    from oracle import oracle as orc
    from rectified_spaattn_amd import _core, _lib
    from test_gpu_parity import TOL, _spec
    meta, _ = load_op_case(name)
    q, k, v, lay, nbr = case_inputs(meta)
    tq, tk, tv = (torch.from_numpy(x).to("cuda:0", dt) for x in (q, k, v))
    q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
    ref = orc.rectified_attention(q, k, v, lay, meta["top_k"], meta["p"], nbr)
    outs = []
    try:
        for w in (1, 0):
            assert _lib.lib().rsa_set_tuning(b"k5_w64", w) == 0
            out = _core.rectified_attention(tq, tk, tv, _spec(lay), meta["top_k"], meta["p"],
                                            torch.from_numpy(nbr) if nbr is not None else None)
            torch.cuda.synchronize()
            outs.append(out.float().cpu().numpy().reshape(ref.shape))
    finally:
        _lib.lib().rsa_set_tuning(b"k5_w64", 3)
    mx, mean = TOL[dt]
    for o in outs:
        err = np.abs(o - ref)
        assert err.max() <= mx and err.mean() <= mean
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    assert np.abs(outs[0] - outs[1]).max() <= ulp * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("name", ["wan_d64_1100", "cogvideo_994"])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_k5_64row_kernel_at_head_dim_64_against_the_32row_kernel(name, dt):
    """Round 6: head dim 64 (CogVideoX; the wan_d64 fixture) runs the 64-rows-per-wave K5 too (bit 1 of k5_w64; 8 + 8 MFMAs per
    sub-step, 4-KiB half-tiles).  Both kernels against the oracle and against each other within one output ulp."""
    from conftest import case_inputs, load_op_case
    from oracle import oracle as orc
    from rectified_spaattn_amd import _core, _lib
    from test_gpu_parity import TOL, _spec
    meta, _ = load_op_case(name)
    assert meta["D"] == 64
    q, k, v, lay, nbr = case_inputs(meta)
    tq, tk, tv = (torch.from_numpy(x).to("cuda:0", dt) for x in (q, k, v))
    q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
    ref = orc.rectified_attention(q, k, v, lay, meta["top_k"], meta["p"], nbr)
    outs = []
    try:
        for w in (3, 1):
            assert _lib.lib().rsa_set_tuning(b"k5_w64", w) == 0
            out = _core.rectified_attention(tq, tk, tv, _spec(lay), meta["top_k"], meta["p"],
                                            torch.from_numpy(nbr) if nbr is not None else None)
            torch.cuda.synchronize()
            outs.append(out.float().cpu().numpy().reshape(ref.shape))
    finally:
        _lib.lib().rsa_set_tuning(b"k5_w64", 3)
    mx, mean = TOL[dt]
    for o in outs:
        err = np.abs(o - ref)
        assert err.max() <= mx and err.mean() <= mean, (err.max(), err.mean())
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    assert np.abs(outs[0] - outs[1]).max() <= 2 * ulp * max(1.0, np.abs(ref).max())



@pytest.mark.parametrize("cfg", [("hunyuan", 2, 115456, 115400, 128, 90), ("wan", 3, 27280, 0, 128, 53),
                                 ("cogvideo", 3, 42466, 226, 64, 82), ("wan", 2, 4000, 0, 128, 4)])
def test_k4_split_form_against_fp64_and_the_chain_form(cfg):
    """K4 (round 6): comp = w . vbar on the 2-byte matrix pipe with the fp32 operands split hi + lo (three MFMAs per product), at
    full-size rows (901 columns, HunyuanVideo), a padded layout, head dim 64 and a short ragged one.  comp is a tolerance-only
    quantity: <= 1e-5 against the oracle (tests/test_gpu_parity.py); here both forms against an fp64 product of the device's own
    w and vbar, and against each other."""
    from rectified_spaattn_amd import _core, _lib
    from bench import gen_qkv
    variant, H, S, aux, D, top_k = cfg
    if variant == "hunyuan":
        spec = _core.LayoutSpec.hunyuan(S, aux)
    elif variant == "cogvideo":
        spec = _core.LayoutSpec.cogvideo(S, aux)
    else:
        spec = _core.LayoutSpec.wan(S, 2)
    q, k, v = gen_qkv(H, 0, S, S, D, torch.device(DEV), seed=31)
    L = _lib.lib()
    comps = {}
    try:
        for form in (1, 0):
            assert L.rsa_set_tuning(b"k4_split", 2 * form) == 0        # (2 = the split form at every grid size)
            call = _core.StagedCall(q, k, v, spec, top_k, 0.05, None, reuse_buffers=False)
            call.select()
            torch.cuda.synchronize()
            comps[form] = call.bufs["comp"].clone()
            if form == 1:
                w, vbar = call.bufs["w"].double(), call.bufs["vbar"].double()
                ref = torch.einsum("hij,hjd->hid", w, vbar[:, : w.shape[-1]])
                mag = float(torch.einsum("hij,hjd->hid", w.abs(), vbar[:, : w.shape[-1]].abs()).max())   # sum |w| |vbar|
    finally:
        L.rsa_set_tuning(b"k4_split", 1)
    # the split form drops terms of 2^-16 per product (bound: 2^-15 sum |w| |vbar|); the chain form is fp32-exact to ~1e-7 relative
    for form, rel in ((1, 3.1e-5), (0, 1e-6)):
        err = float((comps[form].double() - ref).abs().max())
        assert err <= rel * mag + 1e-12 and err <= 2e-6, (form, err, mag)
    assert mag > 0.0


@pytest.mark.parametrize("cfg", [("hunyuan", 2, 115456, 115400, 128, torch.bfloat16), ("wan", 3, 27280, 0, 128, torch.float16),
                                 ("cogvideo", 3, 42466, 226, 64, torch.bfloat16), ("wan", 2, 4000, 0, 128, torch.bfloat16),
                                 ("flux", 2, 66048, 512, 128, torch.bfloat16), ("wan", 5, 128 * 65 - 3, 0, 64, torch.float16)])
def test_k2_dma_form_equals_the_staged_form(cfg):
    """K2 (round 6): the visual tiles' operands through LDS-DMA into a double-buffered tile (one barrier per chunk, swizzled source
    chunks, the two chain steps of a ds_read_b128 picked by lane half) against the register-staged form: the chains are the same
    operand for operand, so every score and every GAPR byte must be EQUAL -- full-size rows, edge tiles (NBv not a multiple of 64),
    text-token tiles beside them, both head dims and dtypes."""
    from rectified_spaattn_amd import _core, _lib
    from bench import gen_qkv
    variant, H, S, aux, D, dt = cfg
    spec = {"hunyuan": lambda: _core.LayoutSpec.hunyuan(S, aux), "cogvideo": lambda: _core.LayoutSpec.cogvideo(S, aux),
            "flux": lambda: _core.LayoutSpec.flux(S, aux), "wan": lambda: _core.LayoutSpec.wan(S, 2)}[variant]()
    q, k, v = (x.to(dt) for x in gen_qkv(H, 0, S, S, D, torch.device(DEV), seed=37))
    L = _lib.lib()
    got = {}
    try:
        for form in (32, 16, 0):
            assert L.rsa_set_tuning(b"k2_dma", form) == 0
            call = _core.StagedCall(q, k, v, spec, 8, 0.1, None, reuse_buffers=False)
            call.bufs["scores"].fill_(float("nan")); call.bufs["unrel"].fill_(7)
            call.select()
            torch.cuda.synchronize()
            got[form] = {n: call.bufs[n].clone() for n in ("scores", "unrel", "bitmask", "counts", "R")}
    finally:
        L.rsa_set_tuning(b"k2_dma", 0)
    for form in (32, 16):
        assert not torch.isnan(got[form]["scores"]).any() and int(got[form]["unrel"].max()) <= 1      # every element was written
        for n in got[form]:
            assert torch.equal(got[form][n].view(torch.uint8), got[0][n].view(torch.uint8)), (form, n)
