"""GPU tests of the producers either side of the path: token permutation and the fused
RMSNorm + RoPE (+ concat placement) kernel, against the PyTorch ops they replace."""
import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_permute_tokens_is_exact_and_invertible(dt):
    from rectified_spaattn_amd import glue
    from rectified_spaattn_amd.utils import jenga_gilbert
    l2h, h2l = jenga_gilbert.gilbert_mapping(4, 12, 16)
    S, C = len(l2h), 256
    x = torch.randn(2, S, C, device=DEV).to(dt)
    order = torch.tensor(h2l, dtype=torch.long, device=DEV)   # scripts keep long CUDA tensors (main_hunyuan.py:42)
    inv = torch.tensor(l2h, dtype=torch.long, device=DEV)
    y = glue.permute_tokens(x, order)
    assert torch.equal(y, x[:, order])
    assert torch.equal(glue.permute_tokens(y, inv), x)
    # strided source (a slice of a wider buffer) and a partial gather
    wide = torch.randn(2, S, 2 * C, device=DEV).to(dt)
    assert torch.equal(glue.permute_tokens(wide[:, :, C:], order[:100]), wide[:, order[:100], C:])
    with pytest.raises(AssertionError):
        glue.permute_tokens(x.float(), order)


def test_build_attention_mask_matches_reference_recipe():
    from rectified_spaattn_amd import glue
    enc = torch.zeros(2, 256, dtype=torch.bool, device=DEV)
    enc[0, :200] = True
    enc[1, :17] = True
    mask, eff = glue.build_attention_mask(1024, enc)
    assert mask.shape == (2, 1, 1, 1280) and eff.tolist() == [1224, 1041]
    assert int(mask[0].sum()) == 1224 and bool(mask[1, 0, 0, 1040]) and not bool(mask[1, 0, 0, 1041])


def _unfused(x, heads, norm, rope, rope_tokens):
    from rectified_spaattn_amd import _operator as op
    q = op.split_heads(x, heads)
    if norm is not None:
        q = norm(q)
    if rope is not None:
        q = torch.cat([op.rotary(q[:, :, :rope_tokens], rope), q[:, :, rope_tokens:]], dim=2)
    return q


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hd", [128, 64])
@torch.no_grad()
def test_fused_norm_rope_matches_torch_ops(dt, hd):
    from rectified_spaattn_amd import glue
    heads, S, n_txt = 3, 515, 77
    x = (torch.randn(2, S, heads * hd, device=DEV) * 1.7).to(dt)
    norm = helpers.RMS(hd).to(DEV, dt)
    cos, sin = (t.to(DEV) for t in helpers.rope_tables(S - n_txt, hd))
    ref = _unfused(x, heads, norm, (cos, sin), S - n_txt).float()
    got = glue.qk_norm_rope(x, heads, glue.norm_params(norm), (cos, sin), S - n_txt).float()
    assert got.shape == ref.shape
    d = (got - ref).abs()
    ulp = 2.0 ** (-7 if dt == torch.bfloat16 else -10)
    assert float(d.max()) <= 2 * ulp * float(ref.abs().max())         # at most a last-bit flip
    assert float((d > 0).float().mean()) < 0.02                        # and only on a small fraction
    # norm only, rope only, neither-weight variants
    got_n = glue.qk_norm_rope(x, heads, glue.norm_params(norm), None, 0).float()
    assert float((got_n - _unfused(x, heads, norm, None, 0).float()).abs().max()) <= 2 * ulp * float(ref.abs().max())
    got_r = glue.qk_norm_rope(x, heads, None, (cos, sin), S - n_txt)
    assert torch.equal(got_r, _unfused(x, heads, None, (cos, sin), S - n_txt))   # pure fp32 mul/add: exact
    # destination slice of a concat buffer
    buf = torch.zeros(2, S + 10, heads, hd, dtype=dt, device=DEV)
    glue.qk_norm_rope(x, heads, glue.norm_params(norm), (cos, sin), S - n_txt, out=buf[:, 10:])
    assert torch.equal(buf[:, 10:].transpose(1, 2).float(), got) and float(buf[:, :10].abs().max()) == 0.0


@torch.no_grad()
@pytest.mark.parametrize("which", ["hunyuan_dual", "hunyuan_single", "flux_dual", "flux_single"])
def test_processors_fused_producer_equals_unfused(which):
    from rectified_spaattn_amd import _operator as op
    from rectified_spaattn_amd.rectified_flux_attn import RectifiedFluxSpaAttnProcessor2_0
    from rectified_spaattn_amd.rectified_hunyuan_attn import RectifiedHunyuanVideoSpaAttnProcessor2_0
    heads, hd = 2, 128
    dim = heads * hd
    dt = torch.bfloat16
    hy = which.startswith("hunyuan")
    dual = which.endswith("dual")
    a = helpers.attn_to(helpers.fake_attn(200, heads, hd, added=dual), DEV, dt)
    n_txt = 256 if hy else 512
    hs = helpers.hidden(200, 20, 1, 1024, dim).to(DEV, dt)
    enc = helpers.hidden(200, 21, 1, n_txt, dim).to(DEV, dt)
    if hy:
        mask = torch.zeros(1, 1, 1, 1024 + n_txt, dtype=torch.bool, device=DEV)
        mask[..., :1024 + 200] = True
        rope = tuple(t.to(DEV) for t in helpers.rope_tables(1024, hd))
        proc = RectifiedHunyuanVideoSpaAttnProcessor2_0("flash", 2, None, 0.3, 0)
        call = lambda: proc(a, hs, enc, mask, rope)
    else:
        rope = tuple(t.to(DEV) for t in helpers.rope_tables(1024 + n_txt, hd))
        proc = RectifiedFluxSpaAttnProcessor2_0("flash", 2, None, 0.3, 40, n_txt)
        if dual:
            call = lambda: proc(a, hs, enc, None, rope)
        else:
            both = torch.cat([hs, enc], 1)
            call = lambda: (proc(a, both, None, None, rope),)
    assert op.FUSED_PRODUCER
    fused = call()
    op.FUSED_PRODUCER = False
    try:
        plain = call()
    finally:
        op.FUSED_PRODUCER = True
    for f, p in zip(fused, plain):
        if f is None:
            continue
        assert f.shape == p.shape
        assert float((f.float() - p.float()).abs().max()) <= 3e-2


@pytest.mark.parametrize("n,dt", [(8 * 1000 + 5, torch.bfloat16), (3, torch.float16), (1 * 4096 * 3072, torch.bfloat16)],
                         ids=["ragged", "tiny", "large"])
def test_rel_l1_reduction(n, dt):
    """rsa_rel_l1 (TeaCache statistic, one pass) vs the reference expression evaluated in float64."""
    from rectified_spaattn_amd.teacache import rel_l1_distance
    g = torch.Generator().manual_seed(n % 1000)
    a = torch.randn(n, generator=g).to(DEV, dt)
    b = (a.float().cpu() + 0.1 * torch.randn(n, generator=g)).to(DEV, dt)
    want = ((a.double() - b.double()).abs().mean() / b.double().abs().mean()).item()
    got = rel_l1_distance(a, b)
    assert abs(got - want) <= 2e-5 * want, (got, want)
    assert rel_l1_distance(a, b) == got            # fixed reduction order: repeatable to the bit


def test_teacache_on_device_skips_and_restores():
    from rectified_spaattn_amd.teacache import TeaCache
    tc = TeaCache.hunyuan(num_steps=10, rel_l1_thresh=0.5)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4096, 256, generator=g).to(DEV, torch.bfloat16)
    hidden = torch.zeros(1, 64, 32, device=DEV, dtype=torch.bfloat16)
    decisions = []
    for step in range(10):
        x = x + 0.002 * torch.randn_like(x)
        if tc.should_compute(x):
            h_in = hidden.clone()
            hidden = hidden + 1.0
            tc.store_residual(hidden, h_in)
            decisions.append(True)
        else:
            hidden = tc.apply_residual(hidden)
            decisions.append(False)
    assert decisions[0] and decisions[-1] and not all(decisions)
    assert float(hidden.float().mean()) == 10.0     # a skipped step replays the cached residual (+1)


def _count_mismatch(a, b):
    """(fraction of elements that differ, largest difference in units of the 2-byte spacing at the tensor's largest
    magnitude -- a rotated output near zero inherits the absolute, not the relative, error of its inputs)"""
    a32, b32 = a.float(), b.float()
    diff = (a32 - b32).abs()
    eps = 2.0 ** -8 if a.dtype == torch.bfloat16 else 2.0 ** -11
    return float((diff > 0).float().mean()), float(diff.max() / (b32.abs().max() * eps))


@pytest.mark.parametrize("shape", [(3, 64), (24, 128), (40, 128), (64, 128)])
@torch.no_grad()
def test_across_heads_producer_row_widths(shape):
    """Every instantiation of the one-wave-per-token kernel (3 / 6 / 10 / 16 chunks per lane): head counts of the Wan
    family (24 = TI2V-5B, 40 = 14B) and the extremes, rotation in the fp64 complex form."""
    from rectified_spaattn_amd import glue, rectified_wan21_attn as w21
    heads, hd = shape
    B, S = 1, 37
    x = (torch.randn(B, S, heads * hd, device=DEV) * 0.9).to(torch.bfloat16)
    norm = helpers.RMS(heads * hd).to(DEV, torch.bfloat16)
    fr = helpers.wan_freqs(S, hd).to(DEV)
    want = w21._complex_rope(norm(x).unflatten(2, (heads, -1)).transpose(1, 2), fr)
    got = glue.norm_rope_across_heads(x, heads, glue.norm_params(norm), fr)
    frac, worst = _count_mismatch(got, want)
    assert frac < 5e-3 and worst <= 2.0, (frac, worst)
    assert torch.equal(glue.norm_rope_across_heads(x, heads, None, fr),
                       w21._complex_rope(x.unflatten(2, (heads, -1)).transpose(1, 2), fr))


def test_across_heads_preconditions():
    from rectified_spaattn_amd import _operator as op
    x = torch.zeros(1, 8, 3 * 96, device=DEV, dtype=torch.bfloat16)
    assert not op.fused_heads_ok(x, 3, (None,), None)            # head_dim 96 does not divide 512
    x = torch.zeros(1, 8, 4 * 128, device=DEV, dtype=torch.bfloat16)
    assert op.fused_heads_ok(x, 4, (None,), None)
    assert not op.fused_heads_ok(x.float(), 4, (None,), None)
    assert not op.fused_heads_ok(x, 4, (None,), helpers.wan_freqs(9, 128).to(DEV))      # table of another length
    assert not op.fused_heads_ok(x, 4, (None,), helpers.wan_freqs(8, 128).to(DEV).to(torch.complex64))
    assert not op.fused_heads_ok(x, 4, (torch.nn.LayerNorm(512).to(DEV),), None)        # not an RMSNorm


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kind", ["wan21_complex", "wan22_cos_sin", "norm_only"])
@torch.no_grad()
def test_across_heads_producer_matches_torch_ops(dt, kind):
    """rsa_norm_rope_heads against the module calls it replaces in the Wan processors: RMSNorm over the full inner dim
    (diffusers' order of roundings), then Wan2.1's fp64 complex rotation / Wan2.2's cos-sin rotation.  The only freedom is
    the summation order of the variance (an fp32 ulp of rsqrt): at most a 2-byte ulp on a small fraction of the elements."""
    from rectified_spaattn_amd import glue, rectified_wan21_attn as w21, rectified_wan22_attn as w22
    heads, hd, B, S = 5, 128, 2, 333
    x = (torch.randn(B, S, heads * hd, device=DEV) * 1.3).to(dt)
    norm = helpers.RMS(heads * hd).to(DEV, dt)
    want = norm(x)
    if kind == "wan21_complex":
        fr = helpers.wan_freqs(S, hd).to(DEV)
        want = w21._complex_rope(want.unflatten(2, (heads, -1)).transpose(1, 2), fr)
        got = glue.norm_rope_across_heads(x, heads, glue.norm_params(norm), fr)
    elif kind == "wan22_cos_sin":
        cs = tuple(t.to(DEV) for t in helpers.wan22_rope(S, hd))
        want = w22._cos_sin_rope(want.unflatten(2, (heads, -1)), *cs).transpose(1, 2)
        got = glue.norm_rope_across_heads(x, heads, glue.norm_params(norm), cs)
    else:
        want = want.unflatten(2, (heads, -1)).transpose(1, 2)
        got = glue.norm_rope_across_heads(x, heads, glue.norm_params(norm), None)
    assert got.shape == want.shape == (B, heads, S, hd)
    frac, worst = _count_mismatch(got, want)
    assert frac < 2e-3 and worst <= 2.0, (frac, worst)
    # rotation alone (no norm) is exact: same products, same roundings
    if kind != "norm_only":
        rot = fr if kind == "wan21_complex" else cs
        ref = (w21._complex_rope(x.unflatten(2, (heads, -1)).transpose(1, 2), rot) if kind == "wan21_complex"
               else w22._cos_sin_rope(x.unflatten(2, (heads, -1)), *rot).transpose(1, 2))
        assert torch.equal(glue.norm_rope_across_heads(x, heads, None, rot), ref)


@pytest.mark.parametrize("family", ["wan21", "wan22"])
@torch.no_grad()
def test_wan_processors_take_the_fused_producer(family):
    """The Wan processors route q / k through the one-pass producer when its preconditions hold, and the output of the
    whole processor stays within a 2-byte ulp-level distance of the unfused module path."""
    from rectified_spaattn_amd import _operator as op
    from rectified_spaattn_amd import rectified_wan21_attn as w21, rectified_wan22_attn as w22
    heads, hd, S = 4, 128, 640
    a = helpers.attn_to(helpers.fake_attn(5, heads, hd, wan=True), DEV, torch.bfloat16)
    hs = helpers.hidden(5, 1, 1, S, heads * hd).to(DEV, torch.bfloat16)
    if family == "wan21":
        proc, rot = w21.RectifiedWanT2VSpaAttnProcessor2_0("flash", 2, None, 0.3, processor_id=0), helpers.wan_freqs(S, hd).to(DEV)
    else:
        a.fused_projections = False
        proc = w22.RectifiedWanTI2VSpaAttnProcessor2_0("flash", 2, None, 0.3, processor_id=0)
        rot = tuple(t.to(DEV) for t in helpers.wan22_rope(S, hd))
    assert op.fused_heads_ok(a.to_q(hs), heads, (a.norm_q,), rot)
    fused = proc(a, hs, None, None, rot)
    op.FUSED_PRODUCER = False
    try:
        plain = proc(a, hs, None, None, rot)
    finally:
        op.FUSED_PRODUCER = True
    err = (fused.float() - plain.float()).abs()
    assert float(err.max()) <= 2e-2 * float(plain.float().abs().max()) and float(err.mean()) <= 1e-3 * float(plain.float().abs().mean() + 1e-6) + 1e-4


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hd", [64, 128])
@torch.no_grad()
def test_fused_layernorm_rope_matches_torch_ops(dt, hd):
    """rsa_qk_layernorm_rope (CogVideoX's qk_norm = LayerNorm over the head dim, RoPE on the visual tokens only) against
    torch.nn.LayerNorm + the rotary op: the mean / variance summation order is the only freedom."""
    from rectified_spaattn_amd import glue
    heads, S, n_txt = 3, 515, 77
    x = (torch.randn(2, S, heads * hd, device=DEV) * 1.7 + 0.3).to(dt)
    ln = torch.nn.LayerNorm(hd, eps=1e-6).to(DEV, dt)
    ln.weight.copy_(torch.linspace(0.7, 1.3, hd).to(dt))
    ln.bias.copy_(torch.linspace(-0.2, 0.2, hd).to(dt))
    cos, sin = (t.to(DEV) for t in helpers.rope_tables(S - n_txt, hd))
    want = _unfused(x, heads, ln, (cos, sin), S - n_txt)
    got = glue.qk_norm_rope(x, heads, glue.layernorm_params(ln), (cos, sin), S - n_txt)
    frac, worst = _count_mismatch(got, want)
    assert got.shape == want.shape and frac < 2e-2 and worst <= 2.0, (frac, worst)
    # without affine parameters
    ln2 = torch.nn.LayerNorm(hd, eps=1e-5, elementwise_affine=False).to(DEV, dt)
    frac, worst = _count_mismatch(glue.qk_norm_rope(x, heads, glue.layernorm_params(ln2), None, 0), _unfused(x, heads, ln2, None, 0))
    assert frac < 2e-2 and worst <= 2.0, (frac, worst)


@torch.no_grad()
def test_cogvideo_processor_takes_the_fused_producer():
    from rectified_spaattn_amd import _operator as op
    from rectified_spaattn_amd import rectified_cogvideo_attn as cog
    heads, hd, S_v, n_txt = 4, 64, 768, 226
    a = helpers.attn_to(helpers.fake_attn(9, heads, hd, added=False), DEV, torch.bfloat16)
    a.norm_q, a.norm_k = (torch.nn.LayerNorm(hd, eps=1e-6).to(DEV, torch.bfloat16) for _ in range(2))
    a.prepare_attention_mask = lambda m, s, b: m
    hs = helpers.hidden(9, 1, 1, S_v, heads * hd).to(DEV, torch.bfloat16)
    enc = helpers.hidden(9, 2, 1, n_txt, heads * hd).to(DEV, torch.bfloat16)
    rot = tuple(t.to(DEV) for t in helpers.rope_tables(S_v, hd))
    proc = cog.RectifiedCogVideoXVideoSpaAttnProcessor2_0("flash", 2, None, 0.3, processor_id=0)
    assert op.fused_qk_ok(a.to_q(torch.cat([hs, enc], 1)), heads, (a.norm_q, a.norm_k), rot)
    f_v, f_t = proc(a, hs, enc, None, rot)
    op.FUSED_PRODUCER = False
    try:
        p_v, p_t = proc(a, hs, enc, None, rot)
    finally:
        op.FUSED_PRODUCER = True
    for f, p in ((f_v, p_v), (f_t, p_t)):
        err = (f.float() - p.float()).abs()
        assert float(err.max()) <= 2e-2 * float(p.float().abs().max()) and float(err.mean()) <= 2e-3 * float(p.float().abs().mean()) + 1e-4
