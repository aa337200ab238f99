"""fp8 (e4m3) K5 on the GPU (BASELINE config 5: fp8 Q/K/V on the CDNA4 fp8 MFMA).

* the producer's block-scaled images, block exponents and K mean equal oracle.fp8_operands byte for byte (integer / byte
  work: bit-exact), whether K1 writes them in its pooling pass or the stand-alone producer does
* the kept lists, R and comp are the 2-byte path's (the mask-selection pass does not see the fp8 images)
* the kernel's output is within max|d| <= 4e-2, mean|d| <= 4e-3 of the fp8-aware oracle (same dequantised e4m3
  operands, P kept exact) for N(0,1)-scale V -- measured 2.3e-2 / 2.5e-3, which is the e4m3 rounding of P
* against the bf16 oracle (un-quantised inputs) the stated tolerance is max|d| <= 1.6e-1, mean|d| <= 1.5e-2: here the
  e4m3 rounding of Q, K, V themselves dominates (the fp8-aware oracle alone is 5e-2..1.4e-1 / 1e-2 away from the bf16
  oracle on this data, so SURVEY 8(d)'s provisional 8e-2 is not reachable by any per-head-scaled e4m3 operand set)
"""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

FP8_MAX_VS_BF16, FP8_MEAN_VS_BF16 = 1.6e-1, 1.5e-2   # measured 1.4e-1 / 1.3e-2; see tests/diag/diag_fp8_scale_granularity.py
FP8_MAX_VS_FP8, FP8_MEAN_VS_FP8 = 4e-2, 4e-3
FP8_MAX_VS_PCODE, FP8_MEAN_VS_PCODE = 2e-2, 4e-4     # oracle with the kernel's own P map (p_form="code"); measured <= 1.2e-2 / 2.0e-4
FP8_ROW_REL_MEDIAN, FP8_ROW_REL_MAX = 0.13, 0.30   # measured: median 0.06-0.12, max 0.08-0.27 (e4m3: 2^-4 relative steps on Q, K, V and P)
FP8_ROW_REL_MAX_D64 = 0.40                        # head dim 64: half as many elements average the rounding in a row; measured 0.32-0.33


def _spec(lay):
    from rectified_spaattn_amd import _core
    return _core.LayoutSpec(lay.S, lay.NB_total, lay.NBv, lay.n_txt, lay.kv_valid, lay.pool_valid,
                            lay.text_end_block, lay.ffb, lay.q_text_valid, lay.kv_text_valid)


CASES = [
    ("wan_ragged", lambda: orc.layout_wan(7 * 128 - 37, 2), 2, 3, 0.3, 1, torch.bfloat16),
    ("wan_tiny", lambda: orc.layout_wan(100, 0), 1, 1, 0.5, -1, torch.bfloat16),
    ("hunyuan", lambda: orc.layout_hunyuan(6 * 128 + 256, 6 * 128 + 200), 2, 2, 0.4, 1, torch.bfloat16),
    ("flux", lambda: orc.layout_flux(9 * 128 + 512, 512), 1, 3, 0.2, 0, torch.float16),
    ("wan_keep_all", lambda: orc.layout_wan(5 * 128, 0), 1, 99, 1.5, -1, torch.bfloat16),
    ("wan_odd_tiles", lambda: orc.layout_wan(4 * 128 + 40, 1), 1, 2, 0.6, 1, torch.bfloat16),
    # head dim 64 (the CogVideoX shape's head dim): its own K1 / block-kernel instances and K5 block
    ("cogvideo_d64", lambda: orc.layout_cogvideo(6 * 128 + 226, 226), 2, 2, 0.3, 1, torch.bfloat16, 64),
    ("wan_ragged_d64", lambda: orc.layout_wan(7 * 128 - 37, 2), 2, 3, 0.3, 1, torch.float16, 64),
    ("hunyuan_d64", lambda: orc.layout_hunyuan(5 * 128 + 256, 5 * 128 + 131), 1, 2, 0.4, -1, torch.bfloat16, 64),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_fp8_operator(case):
    from rectified_spaattn_amd import _core, synth
    name, mk, H, top_k, p, nbw, dt = case[:7]
    D = case[7] if len(case) > 7 else 128
    lay = mk()
    q, k, v = synth.structured_qkv(4242 + len(name), 1, H, lay.S, D, smooth=0.0)
    nbr = synth.banded_neighbors(lay.NBv, nbw) if nbw >= 0 else None
    tq, tk, tv = (torch.from_numpy(x).to(DEV, dt) for x in (q, k, v))
    q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
    tn = torch.from_numpy(nbr) if nbr is not None else None
    out, parts = _core.rectified_attention(tq, tk, tv, _spec(lay), top_k, p, tn, return_parts=True, qkv_fp8=True)
    out_ref, parts_ref = _core.rectified_attention(tq, tk, tv, _spec(lay), top_k, p, tn, return_parts=True)
    # selection pass untouched by the fp8 option
    for n in ("bitmask", "cols", "counts", "R", "comp", "probs"):
        a, b = parts[n], parts_ref[n]
        if n == "cols":  # only the first counts[] entries of a row are defined
            cnt = parts["counts"].cpu().numpy().reshape(-1)
            a2, b2 = a.cpu().numpy().reshape(len(cnt), -1), b.cpu().numpy().reshape(len(cnt), -1)
            assert all(np.array_equal(a2[i, :c], b2[i, :c]) for i, c in enumerate(cnt))
        else:
            assert torch.equal(a, b), n
    # byte-exact operand images
    ref8, sel, ops = orc.rectified_attention_fp8(q, k, v, lay, top_k, p, nbr, want_parts=True)
    assert np.array_equal(parts["kmean"].cpu().numpy(), ops["kmean"]), "K mean"
    assert np.array_equal(parts["exps"].cpu().numpy().astype(np.uint32), ops["exps"]), "block exponents"
    assert np.array_equal(parts["q8"].cpu().numpy(), ops["q8"]), "q8"
    assert np.array_equal(parts["k8"].cpu().numpy(), ops["k8"]), "k8"
    assert np.array_equal(parts["v8t"].cpu().numpy(), ops["v8t"]), "v8t"
    # output
    o = out.float().cpu().numpy()
    e8 = np.abs(o - ref8)
    assert e8.max() <= FP8_MAX_VS_FP8 and e8.mean() <= FP8_MEAN_VS_FP8, f"vs fp8 oracle: {e8.max():.3e} {e8.mean():.3e}"
    # ... and against the oracle that forms P exactly as the kernel does (code map + deferred reference): what is left is
    # fp32 accumulation, the 2-byte rounding of the output and codes that sit on a rounding boundary
    ref8c = orc.rectified_attention_fp8(q, k, v, lay, top_k, p, nbr, p_form="code")
    e8c = np.abs(o - ref8c)
    print(f"{name}: vs exact-P oracle {e8.max():.3e} / {e8.mean():.3e}; vs code-map oracle {e8c.max():.3e} / {e8c.mean():.3e}")
    assert e8c.max() <= FP8_MAX_VS_PCODE and e8c.mean() <= FP8_MEAN_VS_PCODE, f"vs code-map oracle: {e8c.max():.3e} {e8c.mean():.3e}"
    ref16 = orc.rectified_attention(q, k, v, lay, top_k, p, nbr)
    e16 = np.abs(o - ref16)
    # (head dim 64: a row carried by one or two keys reproduces V, whose e4m3 step is 6 % of |v| <= 4: 2.5e-1 there)
    assert e16.max() <= (FP8_MAX_VS_BF16 if D == 128 else 2.5e-1) and e16.mean() <= FP8_MEAN_VS_BF16, \
        f"vs bf16 oracle: {e16.max():.3e} {e16.mean():.3e}"
    assert np.isfinite(o).all()
    # the same distance per query row, RELATIVE to that row's own output (an absolute bound says little where |O| is
    # small): ||O_fp8 - O_bf16||_2 / ||O_bf16||_2 per (row, head)
    S_, HD = o.shape[1], o.shape[2]
    d = (o - ref16).reshape(S_, H, HD // H)
    rel = np.linalg.norm(d, axis=-1) / np.maximum(np.linalg.norm(ref16.reshape(S_, H, HD // H), axis=-1), 1e-6)
    print(f"{name}: fp8 vs bf16 oracle, relative L2 per query row: median {np.median(rel):.3f} p99 {np.quantile(rel, 0.99):.3f} "
          f"max {rel.max():.3f}")
    assert np.median(rel) <= FP8_ROW_REL_MEDIAN and rel.max() <= (FP8_ROW_REL_MAX if D == 128 else FP8_ROW_REL_MAX_D64), \
        (np.median(rel), rel.max())


def test_fp8_onecall_matches_staged():
    from rectified_spaattn_amd import _core, synth
    lay = orc.layout_hunyuan(5 * 128 + 256, 5 * 128 + 131)
    q, k, v = synth.structured_qkv(77, 1, 2, lay.S, 128, smooth=0.0)
    tq, tk, tv = (torch.from_numpy(x).to(DEV, torch.bfloat16) for x in (q, k, v))
    nbr = torch.from_numpy(synth.banded_neighbors(lay.NBv, 1))
    a = _core.rectified_attention(tq, tk, tv, _spec(lay), 2, 0.3, nbr, qkv_fp8=True)
    b, _ = _core.rectified_attention_onecall(tq, tk, tv, _spec(lay), 2, 0.3, nbr, qkv_fp8=True)
    assert torch.equal(a, b)


def test_fp8_pv_onecall_matches_staged():
    """rsa_rectified_attention_fp8pv (one C call, what a non-Python host uses) = the staged calls of the pv form, bit for bit."""
    from rectified_spaattn_amd import _core, synth
    lay = orc.layout_hunyuan(7 * 128 + 256, 7 * 128 + 150)
    q, k, v = synth.structured_qkv(23, 1, 2, lay.S, 128, smooth=0.0)
    tq, tk, tv = (torch.from_numpy(x).to(DEV, torch.bfloat16) for x in (q, k, v))
    nbr = torch.from_numpy(synth.banded_neighbors(lay.NBv, 1))
    staged = _core.rectified_attention(tq, tk, tv, _spec(lay), 3, 0.3, nbr, qkv_fp8="pv")
    one, _ = _core.rectified_attention_onecall(tq, tk, tv, _spec(lay), 3, 0.3, nbr, qkv_fp8="pv")
    assert torch.equal(one.reshape(staged.shape), staged)


def test_fp8_rejects_head_dims_it_has_no_kernel_for():
    from rectified_spaattn_amd import _core
    z = torch.zeros(1, 1, 256, 48, dtype=torch.bfloat16, device=DEV)   # (16 / 32 are served zero-padded; 48 is no head dim of the reference)
    with pytest.raises(AssertionError):
        _core.rectified_attention(z, z, z, _core.LayoutSpec.wan(256, 0), 1, 0.3, None, qkv_fp8=True)


def test_fp8_large_magnitudes_and_zero_tensor():
    """Scales follow the data: x1024 inputs quantise to the same bytes with every V exponent 10 higher; an all-zero V gives
    exponent 0 (byte 127) and comp-only output."""
    from rectified_spaattn_amd import _core, synth
    lay = orc.layout_wan(3 * 128, 0)
    q, k, v = synth.structured_qkv(9, 1, 1, lay.S, 128, smooth=0.0)
    tq, tk, tv = (torch.from_numpy(x).to(DEV, torch.bfloat16) for x in (q, k, v))
    _, p1 = _core.rectified_attention(tq, tk, tv, _spec(lay), 1, 0.3, None, return_parts=True, qkv_fp8=True)
    _, p2 = _core.rectified_attention(tq, tk, tv * 1024, _spec(lay), 1, 0.3, None, return_parts=True, qkv_fp8=True)
    assert torch.equal(p1["v8t"], p2["v8t"])  # power-of-two rescale: identical bytes
    assert torch.equal(((p1["exps"] >> 16) & 0xFF) + 10, (p2["exps"] >> 16) & 0xFF)
    assert torch.equal(p1["exps"] & 0xFFFF, p2["exps"] & 0xFFFF)
    o, p3 = _core.rectified_attention(tq, tk, torch.zeros_like(tv), _spec(lay), 1, 0.3, None, return_parts=True,
                                      qkv_fp8=True)
    assert bool((((p3["exps"] >> 16) & 0xFF) == 127).all())
    assert torch.count_nonzero(o) == 0


def test_fp8_switch_reaches_the_reference_shaped_operator():
    """set_qkv_fp8(True) routes rectified_block_sparse_attention (and so the processors) through the fp8 K5."""
    import rectified_spaattn_amd as rsa
    from rectified_spaattn_amd import _core, synth
    from rectified_spaattn_amd.rectified_wan21_attn import rectified_block_sparse_attention
    lay = orc.layout_wan(6 * 128, 1)
    q, k, v = synth.structured_qkv(31, 1, 2, lay.S, 128, smooth=0.0)
    tq, tk, tv = (torch.from_numpy(x).to(DEV, torch.bfloat16) for x in (q, k, v))
    nbr = torch.from_numpy(synth.banded_neighbors(lay.NBv, 1))
    kw = dict(block_neighbor_list=nbr, p_remain_rates=0.3, first_frame_blocks=1)
    o16 = rectified_block_sparse_attention(tq, tk, tv, None, 2, **kw)
    old = rsa.set_qkv_fp8(True)
    try:
        o8 = rectified_block_sparse_attention(tq, tk, tv, None, 2, **kw)
    finally:
        rsa.set_qkv_fp8(old)
    want = _core.rectified_attention(tq, tk, tv, _spec(lay), 2, 0.3, nbr, qkv_fp8=True)
    assert torch.equal(o8, want) and not torch.equal(o8, o16)
    assert torch.equal(rectified_block_sparse_attention(tq, tk, tv, None, 2, **kw), o16)


def test_fp8_batch_and_strided_views():
    """B = 2 and q/k/v handed over as strided views of one packed [B, S, 3, H, D] projection (no copies)."""
    from rectified_spaattn_amd import _core, synth
    lay = orc.layout_flux(4 * 128 + 256, 256)
    B, H, D = 2, 3, 128
    q, k, v = synth.structured_qkv(2025, B, H, lay.S, D, smooth=0.0)
    packed = torch.empty(B, lay.S, 3, H, D, dtype=torch.bfloat16, device=DEV)
    for i, x in enumerate((q, k, v)):
        packed[:, :, i] = torch.from_numpy(x).to(DEV, torch.bfloat16).permute(0, 2, 1, 3)
    tq, tk, tv = (packed[:, :, i].permute(0, 2, 1, 3) for i in range(3))       # [B, H, S, D] views
    assert not tq.is_contiguous()
    out, parts = _core.rectified_attention(tq, tk, tv, _spec(lay), 2, 0.3, None, return_parts=True, qkv_fp8=True)
    qf, kf, vf = (t.float().cpu().numpy() for t in (tq, tk, tv))
    ref, sel, ops = orc.rectified_attention_fp8(qf, kf, vf, lay, 2, 0.3, None, want_parts=True)
    assert np.array_equal(parts["exps"].cpu().numpy().astype(np.uint32), ops["exps"])
    assert np.array_equal(parts["k8"].cpu().numpy(), ops["k8"]) and np.array_equal(parts["v8t"].cpu().numpy(), ops["v8t"])
    err = np.abs(out.float().cpu().numpy() - ref)
    assert err.max() <= FP8_MAX_VS_FP8 and err.mean() <= FP8_MEAN_VS_FP8, f"{err.max():.3e} {err.mean():.3e}"
    same = _core.rectified_attention(tq.contiguous(), tk.contiguous(), tv.contiguous(), _spec(lay), 2, 0.3, None,
                                     qkv_fp8=True)
    assert torch.equal(out, same)


@pytest.mark.parametrize("mk", [lambda: orc.layout_hunyuan(4 * 128 + 256, 4 * 128 + 77), lambda: orc.layout_wan(300, 0),
                                lambda: orc.layout_flux(3 * 128 + 128, 128)], ids=["hunyuan", "wan", "flux"])
def test_fp8_fused_and_standalone_producers(mk):
    """Fused form (rsa_pool_stats_fp8: K1 writes the images of the blocks it pools, a small launch the text-tail blocks of
    Q and K) and the stand-alone rsa_quantize_fp8 (one pass of its own) against the oracle contract, and against each
    other: bit-identical images, exponents and K mean."""
    from rectified_spaattn_amd import _core, synth
    lay = mk()
    q, k, v = synth.structured_qkv(808, 1, 2, lay.S, 128, smooth=0.0)
    q[0, 1, lay.S - 1, 5] = 37.0          # the largest |q| sits in the last row (text tail / ragged end)
    k[0, 0, min(lay.NBv * 128, lay.S) - 1, 9] = -41.0
    tq, tk, tv = (torch.from_numpy(x).to(DEV, torch.bfloat16) for x in (q, k, v))
    call = _core.StagedCall(tq, tk, tv, _spec(lay), 2, 0.3, None, qkv_fp8=True)
    call.select()
    fused = {n: t.clone() for n, t in call.fp8.items()}
    for t in call.fp8.values():
        t.zero_()
    call.quantize()
    alone = call.fp8
    qf, kf, vf = (t.float().cpu().numpy() for t in (tq, tk, tv))
    want = orc.fp8_operands(qf, kf, vf, lay)
    BH = 2
    for got, tag in ((fused, "fused"), (alone, "stand-alone")):
        ex = _core.fp8_exps(got["scales"], BH, lay.NB_total).cpu().numpy().astype(np.uint32)
        assert np.array_equal(ex, want["exps"]), tag
        assert np.array_equal(_core.fp8_kmean(got["scales"], BH, lay.NB_total).cpu().numpy(), want["kmean"]), tag
        for n in ("q8", "k8", "v8t"):
            assert np.array_equal(got[n].cpu().numpy(), want[n]), f"{tag} {n}"
    # the statistics K1 leaves behind do not depend on whether it also wrote the images
    plain = _core.StagedCall(tq, tk, tv, _spec(lay), 2, 0.3, None)
    plain.select()
    for n in ("qbar", "aq", "kbar", "ak", "vbar", "bitmask", "R", "comp"):
        assert torch.equal(plain.bufs[n], call.bufs[n]), n


def test_fp8_smooth_k_removes_a_common_key_component():
    """Real K tensors carry a large component shared by all tokens; q.(k - mu) only shifts each row's scores, so the
    producer subtracts a mean (of 8 sampled blocks) before quantising.  With a bias of ~6 sigma added to every key the
    smoothed path stays at the unbiased error level while images without it (tuning key fp8_smooth_k = 0) are several
    times worse."""
    from rectified_spaattn_amd import _core, _lib, synth
    lay = orc.layout_wan(6 * 128, 0)
    q, k, v = synth.structured_qkv(515, 1, 2, lay.S, 128, smooth=0.0)
    bias = np.random.default_rng(5).standard_normal(128).astype(np.float32) * 6.0
    kb = k + bias[None, None, None, :]
    tq, tk, tv = (torch.from_numpy(x).to(DEV, torch.bfloat16) for x in (q, kb, v))
    qf, kf, vf = (t.float().cpu().numpy() for t in (tq, tk, tv))
    ref16 = orc.rectified_attention(qf, kf, vf, lay, 99, 1.5, None)          # keep-all: the mask plays no role
    call = _core.StagedCall(tq, tk, tv, _spec(lay), 99, 1.5, None, qkv_fp8=True)
    call.select()
    smooth = call.attend().float().cpu().numpy().reshape(ref16.shape)
    try:
        assert _lib.lib().rsa_set_tuning(b"fp8_smooth_k", 0) == 0
        call.select()
        plain = call.attend().float().cpu().numpy().reshape(ref16.shape)
        assert np.array_equal(call.fp8["k8"].cpu().numpy(), orc.fp8_operands(qf, kf, vf, lay, smooth_k=False)["k8"])
    finally:
        _lib.lib().rsa_set_tuning(b"fp8_smooth_k", 1)
    e_s, e_p = np.abs(smooth - ref16).mean(), np.abs(plain - ref16).mean()
    ref8 = orc.rectified_attention_fp8(qf, kf, vf, lay, 99, 1.5, None)
    assert np.abs(smooth - ref8).max() <= FP8_MAX_VS_FP8
    assert e_s <= FP8_MEAN_VS_BF16 and e_p >= 2.0 * e_s, f"smooth {e_s:.3e} plain {e_p:.3e}"


@pytest.mark.parametrize("D", [128, 64])
@pytest.mark.parametrize("Sq,Sk,qs,ks", [(300, 520, None, None), (384, 384, 256, 200), (129, 1000, 1, 64)],
                         ids=["plain", "two_segment", "ragged"])
def test_fp8_dense_kernel(Sq, Sk, qs, ks, D):
    """rsa_dense_fwd_fp8 (fullattn's device path with e4m3 operands) vs the fp8-aware dense oracle."""
    from rectified_spaattn_amd import _core
    g = torch.Generator().manual_seed(Sq * 7 + Sk)
    H = 2
    q = torch.randn(1, H, Sq, D, generator=g).to(DEV, torch.bfloat16)
    k = torch.randn(1, H, Sk, D, generator=g).to(DEV, torch.bfloat16)
    v = torch.randn(1, H, Sk, D, generator=g).to(DEV, torch.bfloat16)
    out = _core.dense_attention(q, k, v, qs, ks, qkv_fp8=True)          # [1, Sq, H, D]
    ref16 = _core.dense_attention(q, k, v, qs, ks)
    assert torch.isfinite(out.float()).all()
    for h in range(H):
        ref = orc.dense_attention_fp8(*(t[0, h].float().cpu().numpy() for t in (q, k, v)), qs, ks)
        err = np.abs(out[0, :, h].float().cpu().numpy() - ref)
        assert err.max() <= FP8_MAX_VS_FP8 and err.mean() <= FP8_MEAN_VS_FP8, f"{err.max():.3e} {err.mean():.3e}"
    d16 = (out.float() - ref16.float()).abs()
    assert d16.max() <= FP8_MAX_VS_BF16 and d16.mean() <= FP8_MEAN_VS_BF16


def test_fullattn_dense_fp8_switch():
    import rectified_spaattn_amd as rsa
    from rectified_spaattn_amd.attn import fullattn
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(1, 3, 500, 128, generator=g).to(DEV, torch.bfloat16) for _ in range(3))
    o16 = fullattn(q, k, v, mode="torch")
    old = rsa.set_dense_fp8(True)
    try:
        o8 = fullattn(q, k, v, mode="torch")
    finally:
        rsa.set_dense_fp8(old)
    assert o8.shape == o16.shape and not torch.equal(o8, o16)
    assert (o8.float() - o16.float()).abs().max() <= FP8_MAX_VS_BF16
    assert torch.equal(fullattn(q, k, v, mode="torch"), o16)


def test_fp8_hip_graph_capture_and_replay():
    """The fp8 path (K mean, K1 writing the images, tail blocks, fp8 K5) is capturable and replays on new data."""
    from rectified_spaattn_amd import _core, synth
    qa, ka, va = (torch.from_numpy(x).to(DEV, torch.bfloat16) for x in synth.structured_qkv(41, 1, 2, 1024, 128))
    qb, kb, vb = (torch.from_numpy(x * 3.0).to(DEV, torch.bfloat16) for x in synth.structured_qkv(42, 1, 2, 1024, 128))
    spec = _core.LayoutSpec.wan(1024, 1)
    q, k, v = qa.clone(), ka.clone(), va.clone()
    call = _core.StagedCall(q, k, v, spec, 3, 0.3, None, qkv_fp8=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        call.select(); call.attend()   # warm-up outside capture
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        call.select()
        call.attend()
    q.copy_(qb); k.copy_(kb); v.copy_(vb)
    g.replay()
    torch.cuda.synchronize()
    eager = _core.rectified_attention(qb, kb, vb, spec, 3, 0.3, None, qkv_fp8=True)
    assert torch.equal(call.out.view(1, 1024, 256), eager)


def _random_fp8_cases():
    import test_gpu_random_layouts as R
    return R._cases()   # both head dims


@pytest.mark.parametrize("case", _random_fp8_cases(), ids=lambda c: f"{c[0]}-{c[1]}")
def test_fp8_random_layouts(case):
    """The randomised layouts of test_gpu_random_layouts (both head dims) through the fp8 K5: same mask as the 2-byte
    path, byte-exact images, output within the fp8 tolerance of the fp8-aware oracle, no NaN in the awkward corners."""
    from rectified_spaattn_amd import _core, synth
    i, variant, D, H, lay, top_k, p, nbw = case
    q, k, v = synth.structured_qkv(1000 + i, 1, H, lay.S, D, smooth=0.5 if i % 3 == 0 else 0.0)
    nbr = synth.banded_neighbors(lay.NBv, nbw) if nbw >= 0 else None
    dt = torch.bfloat16 if i % 2 == 0 else torch.float16
    tq, tk, tv = (torch.from_numpy(x).to(DEV, dt) for x in (q, k, v))
    q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
    tn = torch.from_numpy(nbr) if nbr is not None else None
    out, parts = _core.rectified_attention(tq, tk, tv, _spec(lay), top_k, p, tn, return_parts=True, qkv_fp8=True)
    ref8, sel, ops = orc.rectified_attention_fp8(q, k, v, lay, top_k, p, nbr, want_parts=True)
    for bh in range(H):
        kept = orc.unpack_bits(parts["bitmask"][bh].cpu().numpy().view(np.uint32), lay.NB_total)
        assert np.array_equal(kept, sel[bh]["kept"]), f"mask (case {i})"
    assert np.array_equal(parts["exps"].cpu().numpy().astype(np.uint32), ops["exps"])
    for n in ("q8", "k8", "v8t"):
        assert np.array_equal(parts[n].cpu().numpy(), ops[n]), n
    o = out.float().cpu().numpy()
    assert np.isfinite(o).all()
    err = np.abs(o - ref8)
    assert err.max() <= 6e-2 and err.mean() <= 6e-3, f"case {i}: max {err.max():.3e} mean {err.mean():.3e}"


@pytest.mark.parametrize("case", _random_fp8_cases(), ids=lambda c: f"{c[0]}-{c[1]}")
def test_fp8_pv_random_layouts(case):
    """The randomised layouts (both head dims) through the pv form: the 2-byte path's mask, the fp8 path's V image, output within the
    pv tolerance of the oracle with the same two choices, bit-identical to the compiled twin, no NaN in the awkward corners."""
    from rectified_spaattn_amd import _core, _lib, synth
    i, variant, D, H, lay, top_k, p, nbw = case
    q, k, v = synth.structured_qkv(2000 + i, 1, H, lay.S, D, smooth=0.5 if i % 3 == 0 else 0.0)
    nbr = synth.banded_neighbors(lay.NBv, nbw) if nbw >= 0 else None
    dt = torch.bfloat16 if i % 2 == 0 else torch.float16
    tq, tk, tv = (torch.from_numpy(x).to(DEV, dt) for x in (q, k, v))
    q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
    tn = torch.from_numpy(nbr) if nbr is not None else None
    out, parts = _core.rectified_attention(tq, tk, tv, _spec(lay), top_k, p, tn, return_parts=True, qkv_fp8="pv")
    try:
        assert _lib.lib().rsa_set_tuning(b"fp8_variant", 1) == 0
        twin = _core.rectified_attention(tq, tk, tv, _spec(lay), top_k, p, tn, qkv_fp8="pv")
    finally:
        _lib.lib().rsa_set_tuning(b"fp8_variant", 0)
    assert torch.equal(out, twin)
    refc, sel, ops = orc.rectified_attention_fp8(q, k, v, lay, top_k, p, nbr, want_parts=True, p_form="code", qk="2byte")
    for bh in range(H):
        kept = orc.unpack_bits(parts["bitmask"][bh].cpu().numpy().view(np.uint32), lay.NB_total)
        assert np.array_equal(kept, sel[bh]["kept"]), f"mask (case {i})"
    assert np.array_equal(parts["v8t"].cpu().numpy(), ops["v8t"])
    o = out.float().cpu().numpy()
    assert np.isfinite(o).all()
    err = np.abs(o - refc)
    # (max: a score on a rounding boundary of the code map moves one P by a whole e4m3 step; with few kept keys -- these layouts keep
    # two to five blocks -- one such step shows in the output: 5e-2 here against 3e-2 on the structured cases; the all-e4m3 form's
    # random-layout bound is 6e-2)
    # mean: some of these layouts are a few dozen rows (1.1e-3 on 29 rows at head dim 64): twice the structured cases' bound
    assert err.max() <= 5e-2 and err.mean() <= 2 * PV_MEAN_VS_ORACLE, f"case {i}: max {err.max():.3e} mean {err.mean():.3e}"


def test_fp8_dense_smooth_k():
    """Dense fp8 with a common component of 8 sigma on every key: stays at the unbiased error level."""
    from rectified_spaattn_amd import _core
    g = torch.Generator().manual_seed(11)
    q = torch.randn(1, 2, 700, 128, generator=g)
    k = torch.randn(1, 2, 2300, 128, generator=g)
    v = torch.randn(1, 2, 2300, 128, generator=g)
    bias = torch.randn(128, generator=g) * 8.0
    errs = []
    for kk in (k, k + bias):
        tq, tk, tv = (x.to(DEV, torch.bfloat16) for x in (q, kk, v))
        o8 = _core.dense_attention(tq, tk, tv, qkv_fp8=True).float()
        o16 = _core.dense_attention(tq, tk, tv).float()
        errs.append((o8 - o16).abs().mean().item())
        ref = orc.dense_attention_fp8(*(t[0, 1].float().cpu().numpy() for t in (tq, tk, tv)))
        e = np.abs(o8[0, :, 1].cpu().numpy() - ref)
        assert e.max() <= FP8_MAX_VS_FP8 and e.mean() <= FP8_MEAN_VS_FP8, f"{e.max():.3e} {e.mean():.3e}"
    assert errs[1] <= 1.5 * errs[0] + 1e-3, errs


def test_fp8_p_forms_code_map_against_exact_exponential():
    """The product forms P through the e4m3 code map (one conversion per score); tuning key fp8_variant also reaches its
    compiled twin (1: must be bit-identical), the block with the staging behind the barrier (3: bit-identical) and the exact-exponential form (2: v_exp_f32 + round-to-nearest e4m3).  Both
    forms sit inside the fp8 tolerance of the exact-P oracle, the code map within 1.35x of the exponential form's mean error
    (simulation: 1.2x, tests/diag/diag_fp8_pmap.py), and against the bf16 oracle -- where the e4m3 rounding of Q, K, V
    dominates -- the two are indistinguishable (<= 3 %)."""
    from rectified_spaattn_amd import _core, _lib, synth
    lay = orc.layout_hunyuan(6 * 128 + 256, 6 * 128 + 200)
    H, top_k, p = 2, 3, 0.4
    q, k, v = synth.structured_qkv(977, 1, H, lay.S, 128, smooth=0.0)
    tq, tk, tv = (torch.from_numpy(x).to(DEV, torch.bfloat16) for x in (q, k, v))
    q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
    outs = {}
    try:
        for var in (0, 1, 2, 3):
            assert _lib.lib().rsa_set_tuning(b"fp8_variant", var) == 0
            outs[var] = _core.rectified_attention(tq, tk, tv, _spec(lay), top_k, p, None, qkv_fp8=True).float().cpu().numpy()
    finally:
        _lib.lib().rsa_set_tuning(b"fp8_variant", 0)
    assert np.array_equal(outs[0], outs[1]), "hand-placed block and its compiled twin must agree bit for bit"
    assert np.array_equal(outs[0], outs[3]), "staging inside the block (product) and behind the barrier (3) must agree bit for bit"
    ref8 = orc.rectified_attention_fp8(q, k, v, lay, top_k, p, None)
    ref16 = orc.rectified_attention(q, k, v, lay, top_k, p, None)
    e_code, e_exp = np.abs(outs[0] - ref8), np.abs(outs[2] - ref8)
    print(f"vs exact-P oracle: code map {e_code.max():.3e} / {e_code.mean():.3e}, exponential {e_exp.max():.3e} / {e_exp.mean():.3e}")
    for e in (e_code, e_exp):
        assert e.max() <= FP8_MAX_VS_FP8 and e.mean() <= FP8_MEAN_VS_FP8
    assert e_code.mean() <= 1.35 * e_exp.mean()
    b_code, b_exp = np.abs(outs[0] - ref16).mean(), np.abs(outs[2] - ref16).mean()
    print(f"vs bf16 oracle (mean): code map {b_code:.4e}, exponential {b_exp:.4e}")
    assert b_code <= 1.03 * b_exp


@pytest.mark.parametrize("D", [128, 64])
@pytest.mark.parametrize("Sq,Sk,qs,ks", [(700, 700, None, None), (300, 1000, None, None), (1000, 300, None, None),
                                         (640, 640, 200, 260)], ids=["square", "more_keys", "fewer_keys", "two_segments"])
def test_fp8_dense_kernel_causal(Sq, Sk, qs, ks, D):
    """The e4m3 dense kernel's causal form (bottom-right aligned inside each segment, as rsa_dense_causal_fwd): rows that
    see no key give zeros; against the fp8-aware oracle and, loosely, the 2-byte kernel."""
    from rectified_spaattn_amd import _core
    g = torch.Generator().manual_seed(Sq + 3 * Sk + D)
    H = 2
    q = torch.randn(1, H, Sq, D, generator=g).to(DEV, torch.bfloat16)
    k = torch.randn(1, H, Sk, D, generator=g).to(DEV, torch.bfloat16)
    v = torch.randn(1, H, Sk, D, generator=g).to(DEV, torch.bfloat16)
    out = _core.dense_attention(q, k, v, qs, ks, qkv_fp8=True, causal=True)
    ref16 = _core.dense_attention(q, k, v, qs, ks, causal=True)
    assert torch.isfinite(out.float()).all()
    for h in range(H):
        ref = orc.dense_attention_fp8(*(t[0, h].float().cpu().numpy() for t in (q, k, v)), qs, ks, causal=True)
        got = out[0, :, h].float().cpu().numpy()
        rel = np.linalg.norm(got - ref, axis=-1) / np.maximum(np.linalg.norm(ref, axis=-1), 1e-3)
        assert rel.max() <= 0.2 and np.median(rel) <= 0.05, (rel.max(), np.median(rel))   # rows with 1-2 keys reproduce V
    d16 = (out.float() - ref16.float()).abs()
    assert d16.mean() <= FP8_MEAN_VS_BF16


# ---- the "pv" form (round 5): 2-byte Q . K^T, e4m3 only for P . V (rsa_block_sparse_fwd_fp8pv) --------------------------------
PV_MAX_VS_ORACLE, PV_MEAN_VS_ORACLE = 3e-2, 1e-3       # oracle with the kernel's own P map and the 2-byte scores; the kernel also
                                                       # rounds its scaled q to the 2-byte type, as the 2-byte kernels do
PV_MAX_VS_BF16, PV_MEAN_VS_BF16, PV_REL_L1 = 8e-2, 8e-3, 0.06   # SURVEY 8(d)'s bound for fp8 operands: this form meets it


@pytest.mark.parametrize("case", [c for c in CASES if len(c) == 7], ids=lambda c: c[0])
def test_fp8_pv_form(case):
    """Scores from the 2-byte q and k, e4m3 P and V: the selection pass and the V image are the fp8 path's (byte for byte), the
    output is the oracle's with the same two choices (P by the code map: p_form="code", qk="2byte") within fp32 accumulation
    and the 2-byte rounding of the scaled q, and sits within SURVEY 8(d)'s 8e-2 of the bf16 oracle -- which the all-e4m3 form
    does not (1.4e-1)."""
    from rectified_spaattn_amd import _core, synth
    name, mk, H, top_k, p, nbw, dt = case
    lay = mk()
    q, k, v = synth.structured_qkv(4242 + len(name), 1, H, lay.S, 128, smooth=0.0)
    nbr = synth.banded_neighbors(lay.NBv, nbw) if nbw >= 0 else None
    tq, tk, tv = (torch.from_numpy(x).to(DEV, dt) for x in (q, k, v))
    q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
    tn = torch.from_numpy(nbr) if nbr is not None else None
    out, parts = _core.rectified_attention(tq, tk, tv, _spec(lay), top_k, p, tn, return_parts=True, qkv_fp8="pv")
    out8, parts8 = _core.rectified_attention(tq, tk, tv, _spec(lay), top_k, p, tn, return_parts=True, qkv_fp8=True)
    for n in ("bitmask", "counts", "R", "comp", "v8t"):
        assert torch.equal(parts[n], parts8[n]), n
    # the pv form's producer writes the V image and the V exponents only (no Q / K images, no K mean)
    assert torch.equal((parts["exps"] >> 16) & 0xFF, (parts8["exps"] >> 16) & 0xFF) and parts["q8"].numel() == 0 and parts["k8"].numel() == 0
    o = out.float().cpu().numpy()
    assert np.isfinite(o).all()
    refc = orc.rectified_attention_fp8(q, k, v, lay, top_k, p, nbr, p_form="code", qk="2byte")
    ec = np.abs(o - refc)
    ref16 = orc.rectified_attention(q, k, v, lay, top_k, p, nbr)
    e16 = np.abs(o - ref16)
    e8 = np.abs(out8.float().cpu().numpy() - ref16)
    rel = e16.sum() / np.abs(ref16).sum()
    print(f"{name}: pv vs its oracle {ec.max():.3e} / {ec.mean():.3e}; vs bf16 oracle {e16.max():.3e} / {e16.mean():.3e} rel-L1 {rel:.4f} "
          f"(all-e4m3: {e8.max():.3e} / {e8.mean():.3e} rel-L1 {e8.sum() / np.abs(ref16).sum():.4f})")
    assert ec.max() <= PV_MAX_VS_ORACLE and ec.mean() <= PV_MEAN_VS_ORACLE, f"vs pv oracle: {ec.max():.3e} {ec.mean():.3e}"
    assert e16.max() <= PV_MAX_VS_BF16 and e16.mean() <= PV_MEAN_VS_BF16 and rel <= PV_REL_L1, (e16.max(), e16.mean(), rel)
    if name not in ("wan_keep_all", "wan_tiny"):
        assert e16.mean() < 0.6 * e8.mean(), "the pv form is meant to be clearly closer to the bf16 oracle than the all-e4m3 form"


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_fp8_pv_hand_placed_block_against_its_compiled_twin(dt):
    """The pv kernel's tile block is one hand-placed instruction stream (gen_k5_block.py::gen_block8h), six tiles per loop trip;
    the product's also issues the wave's LDS-DMA pieces itself (for every tile: past the end of the walk the last tile again).
    Tuning key fp8_variant 1 launches the same arithmetic as hipcc schedules it, 3 the hand-placed block with the staging behind
    the barrier.  Bit for bit equal, for kept lists of every
    length 1 .. 14 blocks (2 .. 28 tiles: every remainder of the six-tile trip, with and without the half tile at the end of the
    valid keys) and with text rows."""
    from rectified_spaattn_amd import _core, _lib, synth
    L = _lib.lib()
    lays = [("wan", orc.layout_wan(14 * 128 - 40, 1)), ("hunyuan", orc.layout_hunyuan(9 * 128 + 256, 9 * 128 + 130))]
    for name, lay in lays:
        q, k, v = synth.structured_qkv(31 + len(name), 1, 2, lay.S, 128, smooth=0.0)
        tq, tk, tv = (torch.from_numpy(x).to(DEV, dt) for x in (q, k, v))
        for top_k in range(1, lay.NBv + 1):
            outs = {}
            try:
                for var in (0, 1, 3):
                    assert L.rsa_set_tuning(b"fp8_variant", var) == 0
                    outs[var] = _core.rectified_attention(tq, tk, tv, _spec(lay), top_k, 0.0, None, qkv_fp8="pv")
            finally:
                L.rsa_set_tuning(b"fp8_variant", 0)
            assert torch.isfinite(outs[0].float()).all()
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[3]), (name, top_k)


def test_fp8_pv_kept_lists_longer_than_the_lds_window():
    """The pv kernel keeps a 1 024-entry window of the kept list in LDS and refills it half a window at a time while it walks
    (with the whole list of a long head its 72 KiB of rings would leave one workgroup per CU).  2 200 key blocks, 2 150 kept per
    query block: two refills per walk.  Checked against the 2-byte kernel on the same kept lists (relative L1 and max inside the
    pv bounds) and bit for bit against the compiled twin."""
    from rectified_spaattn_amd import _core, _lib
    L = _lib.lib()
    NB, top_k = 2200, 2150
    S = NB * 128 - 77
    g = torch.Generator(device="cpu").manual_seed(99)
    q, k, v = (torch.randn(1, 1, S, 128, generator=g).to(torch.bfloat16).to(DEV) for _ in range(3))
    spec = _core.LayoutSpec.wan(S, 0)
    ref = _core.StagedCall(q, k, v, spec, top_k, 0.0, None)
    ref.select()
    ref.attend()
    assert int(ref.bufs["counts"].min()) > 2048        # every walk crosses the window twice
    o16 = ref.out.float().clone()
    outs = {}
    try:
        for var in (0, 1):
            assert L.rsa_set_tuning(b"fp8_variant", var) == 0
            c = _core.StagedCall(q, k, v, spec, top_k, 0.0, None, qkv_fp8="pv")
            c.select()
            assert torch.equal(c.bufs["bitmask"], ref.bufs["bitmask"]) and torch.equal(c.bufs["counts"], ref.bufs["counts"])
            c.attend()
            outs[var] = c.out.float().clone()
            del c
    finally:
        L.rsa_set_tuning(b"fp8_variant", 0)
    assert torch.equal(outs[0], outs[1])
    d = (outs[0] - o16).abs()
    rel = float(d.sum() / o16.abs().sum())
    print(f"pv vs 2-byte kernel at {NB} key blocks: max {float(d.max()):.3e} rel-L1 {rel:.4f}")
    assert torch.isfinite(outs[0]).all() and float(d.max()) <= PV_MAX_VS_BF16 and rel <= PV_REL_L1


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("Sq,Sk,qs,ks,causal", [(300, 520, None, None, False), (384, 384, 256, 200, False), (129, 1000, 1, 64, False),
                                                (700, 700, None, None, True), (640, 640, 200, 260, True)],
                         ids=["plain", "two_segment", "ragged", "causal", "causal_two_segments"])
def test_fp8_pv_dense_kernel(Sq, Sk, qs, ks, causal, dt):
    """rsa_dense_fwd_fp8pv (fullattn's device path in the pv form: 2-byte scores, e4m3 P and V) against the dense oracle with the same
    two choices (qk="2byte"), against the 2-byte dense kernel, and bit for bit against the compiled twin of its tile block."""
    from rectified_spaattn_amd import _core, _lib
    g = torch.Generator().manual_seed(Sq * 5 + Sk)
    H = 2
    q = torch.randn(1, H, Sq, 128, generator=g).to(DEV, dt)
    k = torch.randn(1, H, Sk, 128, generator=g).to(DEV, dt)
    v = torch.randn(1, H, Sk, 128, generator=g).to(DEV, dt)
    outs = {}
    try:
        for var in (0, 1):
            assert _lib.lib().rsa_set_tuning(b"fp8_variant", var) == 0
            outs[var] = _core.dense_attention(q, k, v, qs, ks, qkv_fp8="pv", causal=causal)          # [1, Sq, H, D]
    finally:
        _lib.lib().rsa_set_tuning(b"fp8_variant", 0)
    out = outs[0]
    assert torch.equal(outs[0], outs[1])
    ref16 = _core.dense_attention(q, k, v, qs, ks, causal=causal)
    o8 = _core.dense_attention(q, k, v, qs, ks, qkv_fp8=True, causal=causal)
    assert torch.isfinite(out.float()).all()
    for h in range(H):
        ref = orc.dense_attention_fp8(*(t[0, h].float().cpu().numpy() for t in (q, k, v)), qs, ks, causal=causal, qk="2byte")
        err = np.abs(out[0, :, h].float().cpu().numpy() - ref)
        # (rows with one or two keys reproduce V: there the e4m3 image of V is the whole error, 2^-4 relative)
        lim = 0.3 if causal else PV_MAX_VS_ORACLE
        assert err.max() <= lim and err.mean() <= 5 * PV_MEAN_VS_ORACLE, f"{err.max():.3e} {err.mean():.3e}"
    d16 = (out.float() - ref16.float()).abs()
    e16 = (o8.float() - ref16.float()).abs()
    print(f"pv dense vs 2-byte dense: max {float(d16.max()):.3e} mean {float(d16.mean()):.3e} (all-e4m3: {float(e16.max()):.3e} / {float(e16.mean()):.3e})")
    assert d16.mean() <= PV_MEAN_VS_BF16 and d16.mean() < 0.75 * e16.mean()


def test_fullattn_dense_fp8_pv_switch():
    import rectified_spaattn_amd as rsa
    from rectified_spaattn_amd.attn import fullattn
    g = torch.Generator().manual_seed(4)
    q, k, v = (torch.randn(1, 3, 500, 128, generator=g).to(DEV, torch.bfloat16) for _ in range(3))
    o16 = fullattn(q, k, v, mode="torch")
    old = rsa.set_dense_fp8("pv")
    try:
        opv = fullattn(q, k, v, mode="torch")
        q6, k6, v6 = (t[..., :64].contiguous() for t in (q, k, v))
        o6 = fullattn(q6, k6, v6, mode="torch")            # head dim 64 has the form too
    finally:
        rsa.set_dense_fp8(old)
    assert opv.shape == o16.shape and not torch.equal(opv, o16)
    assert (opv.float() - o16.float()).abs().max() <= PV_MAX_VS_BF16
    o6_16 = fullattn(q6, k6, v6, mode="torch")
    assert not torch.equal(o6, o6_16) and (o6.float() - o6_16.float()).abs().max() <= PV_MAX_VS_BF16
    with pytest.raises(ValueError):
        rsa.set_dense_fp8("qk")


def test_fp8_pv_form_public_switch():
    """set_qkv_fp8("pv") reaches the operators; anything but False / True / "pv" is refused."""
    import rectified_spaattn_amd as rsa
    from rectified_spaattn_amd import _core, rectified_wan21_attn as rw, synth
    S = 6 * 128
    q, k, v = synth.structured_qkv(5, 1, 2, S, 128, smooth=0.0)
    tq, tk, tv = (torch.from_numpy(x).to(DEV, torch.bfloat16) for x in (q, k, v))
    spec = _core.LayoutSpec.wan(S, 1)
    want = _core.rectified_attention(tq, tk, tv, spec, 2, 0.3, None, qkv_fp8="pv")
    old = rsa.set_qkv_fp8("pv")
    try:
        got = rw.rectified_block_sparse_attention(tq, tk, tv, None, 2, block_neighbor_list=None, p_remain_rates=0.3, first_frame_blocks=1)
    finally:
        rsa.set_qkv_fp8(old)
    assert torch.equal(got.reshape(want.shape), want)
    with pytest.raises(ValueError):
        rsa.set_qkv_fp8("qk")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_fp8_pv_form_head_dim_64(dt):
    """The pv form at head dim 64 (CogVideoX's): K tiles of 64 keys x 128 bytes, eight v_mfma_f32_32x32x16 per tile, two P . V MFMAs.
    Against the oracle with the same two choices, against the bf16 oracle (inside 8e-2, clearly closer than the all-e4m3 form), bit for
    bit against the compiled twin for kept lists of every length, and the dense kernel in the same form."""
    from rectified_spaattn_amd import _core, _lib, synth
    L = _lib.lib()
    lay = orc.layout_cogvideo(9 * 128 - 30, 226) if hasattr(orc, "layout_cogvideo") else orc.layout_wan(9 * 128 - 30, 1)
    H = 2
    q, k, v = synth.structured_qkv(77, 1, H, lay.S, 64, smooth=0.0)
    tq, tk, tv = (torch.from_numpy(x).to(DEV, dt) for x in (q, k, v))
    q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))
    for top_k in range(1, lay.NBv + 1):
        outs = {}
        try:
            for var in (0, 1):
                assert L.rsa_set_tuning(b"fp8_variant", var) == 0
                outs[var] = _core.rectified_attention(tq, tk, tv, _spec(lay), top_k, 0.0, None, qkv_fp8="pv")
        finally:
            L.rsa_set_tuning(b"fp8_variant", 0)
        assert torch.equal(outs[0], outs[1]), top_k
        # the all-e4m3 kernel at this head dim: product (staging inside the block, ones operand in registers) = compiled twin (1) =
        # hand-placed block with the staging behind the barrier (3), bit for bit
        outs = {}
        try:
            for var in (0, 1, 3):
                assert L.rsa_set_tuning(b"fp8_variant", var) == 0
                outs[var] = _core.rectified_attention(tq, tk, tv, _spec(lay), top_k, 0.0, None, qkv_fp8=True)
        finally:
            L.rsa_set_tuning(b"fp8_variant", 0)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[3]), top_k
    top_k, p = 3, 0.3
    out, parts = _core.rectified_attention(tq, tk, tv, _spec(lay), top_k, p, None, return_parts=True, qkv_fp8="pv")
    out8 = _core.rectified_attention(tq, tk, tv, _spec(lay), top_k, p, None, qkv_fp8=True)
    o = out.float().cpu().numpy()
    refc = orc.rectified_attention_fp8(q, k, v, lay, top_k, p, None, p_form="code", qk="2byte")
    ref16 = orc.rectified_attention(q, k, v, lay, top_k, p, None)
    ec, e16, e8 = np.abs(o - refc), np.abs(o - ref16), np.abs(out8.float().cpu().numpy() - ref16)
    print(f"D=64 pv vs its oracle {ec.max():.3e} / {ec.mean():.3e}; vs bf16 oracle {e16.max():.3e} / {e16.mean():.3e} (all-e4m3 {e8.max():.3e} / {e8.mean():.3e})")
    assert ec.max() <= PV_MAX_VS_ORACLE and ec.mean() <= PV_MEAN_VS_ORACLE
    assert e16.max() <= PV_MAX_VS_BF16 and e16.mean() <= PV_MEAN_VS_BF16 and e16.mean() < 0.75 * e8.mean()
    # dense kernel, same form
    g = torch.Generator().manual_seed(5)
    dq, dk, dv = (torch.randn(1, 2, n, 64, generator=g).to(DEV, dt) for n in (300, 520, 520))
    od = _core.dense_attention(dq, dk, dv, qkv_fp8="pv")
    rd = _core.dense_attention(dq, dk, dv)
    for h in range(2):
        ref = orc.dense_attention_fp8(*(t[0, h].float().cpu().numpy() for t in (dq, dk, dv)), qk="2byte")
        err = np.abs(od[0, :, h].float().cpu().numpy() - ref)
        assert err.max() <= PV_MAX_VS_ORACLE and err.mean() <= 5 * PV_MEAN_VS_ORACLE, (err.max(), err.mean())
    assert (od.float() - rd.float()).abs().mean() <= PV_MEAN_VS_BF16


@pytest.mark.parametrize("mode", [True, "pv"], ids=["e4m3", "pv"])
def test_fp8_config5_full_size_ragged_last_block_and_the_bench_check(mode):
    """BASELINE config 5 at its own size (Wan2.2-TI2V 720p 121f: S = 27 280 = 213 x 128 + 16, top_k 53), fp8 operands.  The last query
    block has 16 rows that hang on a handful of keys; there the e4m3 rounding of Q and K alone moves O by 0.35 (the oracle on the
    dequantised operands against the oracle on the inputs: tests/diag/diag_fp8_ragged_block.py), which is the number format and not
    the kernel.  So the kernel is held to the oracle on ITS operands (P formed as the kernel forms it, and P exact), the format's
    distance is bounded only in the mean, and `bench.check_output` -- the check the fp8 bench lines carry -- must say the same."""
    import bench
    from rectified_spaattn_amd import _core
    wl = bench.WORKLOADS["wan22_ti2v_720p_121f"]
    spec = bench.make_spec(wl)
    H = 3
    tq, tk, tv = bench.gen_inputs(wl, H, 0, torch.device(DEV), "iid")
    S = tq.shape[2]
    lay = orc.layout_wan(S, wl["ffb"])
    call = _core.StagedCall(tq, tk, tv, spec, wl["top_k"], 0.0, None, qkv_fp8=mode, reuse_buffers=False)
    call.select(); call.attend()
    torch.cuda.synchronize()
    chk = bench.check_output(call, spec, mode)
    assert chk["ok"] and chk["finite"], chk
    assert chk["worst_at"] is not None and chk["vs_unquantised"]["mean_abs"] <= chk["vs_unquantised"]["tol_mean_abs"]
    # the oracle on head 0: first, middle, the last full and the ragged last query block
    h = 0
    qh, kh, vh = (x[0, h].float().cpu().numpy() for x in (call.q, call.k, call.v))
    q8, k8, v8, ops = orc.fp8_dequantized_qkv(qh[None, None], kh[None, None], vh[None, None], lay)
    q8, k8, v8 = q8[0, 0], k8[0, 0], v8[0, 0]
    SP = spec.NB_total * 128
    assert np.array_equal(call.fp8["v8t"].view(H, SP // 64, 128, 64)[h].cpu().numpy(), ops["v8t"][0]), "v8t"
    if mode is True:
        assert np.array_equal(call.fp8["q8"].view(H, SP, 128)[h].cpu().numpy(), ops["q8"][0]), "q8"
        assert np.array_equal(call.fp8["k8"].view(H, SP, 128)[h].cpu().numpy(), ops["k8"][0]), "k8"
    else:
        q8, k8 = qh, kh          # pv: the scores come from the 2-byte q and k themselves
    kept = _core.unpack_bitmask(call.bufs["bitmask"][h:h + 1], spec.NB_total)[0].cpu().numpy()
    R, comp = call.bufs["R"][h].cpu().numpy(), call.bufs["comp"][h].cpu().numpy()
    blocks = [0, spec.NBv // 2, spec.NBv - 2, spec.NBv - 1]
    km = kept[blocks].astype(np.uint8)
    fin = lambda o: o * R[blocks][:, None, None] + comp[blocks][:, None, :]   # noqa: E731
    ref8 = fin(orc.sparse_attention_head(q8, k8, v8, lay, km, blocks))
    ref8c = fin(orc.sparse_attention_head_pcode(q8, k8, v8, lay, km, blocks))
    ref16 = fin(orc.sparse_attention_head(qh, kh, vh, lay, km, blocks))
    for a, i in enumerate(blocks):
        nrow = min(128, S - i * 128)
        got = call.out[0, i * 128:i * 128 + nrow, h].float().cpu().numpy()
        d8, d8c, d16 = (np.abs(got - r[a][:nrow]) for r in (ref8, ref8c, ref16))
        fmt = np.abs(ref8[a][:nrow] - ref16[a][:nrow])
        print(f"{mode} block {i} ({nrow} rows): vs exact-P {d8.max():.4f}/{d8.mean():.5f}, vs code-map {d8c.max():.4f}/{d8c.mean():.5f}, "
              f"vs un-quantised {d16.max():.4f}/{d16.mean():.5f} (format alone {fmt.max():.4f}/{fmt.mean():.5f})")
        # (the pv kernel rounds q * sm_scale * log2(e) * 8 to the 2-byte type before its Q.K^T, as the 2-byte kernels round their scaled q:
        # codes on a rounding boundary move, so its code-map distance gets the exact-P bound)
        assert d8c.max() <= (FP8_MAX_VS_PCODE if mode is True else FP8_MAX_VS_FP8), (i, d8c.max())
        assert d8c.mean() <= (FP8_MEAN_VS_PCODE if mode is True else FP8_MEAN_VS_FP8) * (2.0 if nrow < 128 else 1.0), (i, d8c.mean())
        # P exact: the e4m3 rounding of P; a 16-row block whose rows hang on few keys averages less of it (measured 3.0e-2 / 6.1e-3)
        assert d8.max() <= FP8_MAX_VS_FP8 and d8.mean() <= FP8_MEAN_VS_FP8 * (2.0 if nrow < 128 else 1.0), (i, d8.max(), d8.mean())
        # the kernel adds nothing to the format's own distance
        assert d16.max() <= fmt.max() + FP8_MAX_VS_FP8, (i, d16.max(), fmt.max())
