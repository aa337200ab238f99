"""Processor plumbing on CPU (BASELINE config 1: dense CPU attn_processor path, no GPU): this package's
processors against vectors produced by the reference's processors (tests/golden/processors.npz)."""
import os
import types

import numpy as np
import pytest
import torch

import helpers
from conftest import GOLDEN
from rectified_spaattn_amd import attn_processor
from rectified_spaattn_amd.rectified_cogvideo_attn import RectifiedCogVideoXVideoSpaAttnProcessor2_0
from rectified_spaattn_amd.rectified_flux_attn import RectifiedFluxSpaAttnProcessor2_0
from rectified_spaattn_amd.rectified_hunyuan_attn import RectifiedHunyuanVideoSpaAttnProcessor2_0
from rectified_spaattn_amd.rectified_wan21_attn import RectifiedWanT2VSpaAttnProcessor2_0

G = np.load(os.path.join(GOLDEN, "processors.npz"))
heads, hd = 2, 128
dim = heads * hd


def close(a, name, atol=3e-3):
    np.testing.assert_allclose(a.detach().numpy(), G[name].astype(np.float32), atol=atol, rtol=2e-3)


@torch.no_grad()
def test_hunyuan_dual_and_single_stream():
    hs, enc = helpers.hidden(101, 20, 1, 1024, dim), helpers.hidden(101, 21, 1, 256, dim)
    mask = torch.zeros(1, 1, 1, 1280, dtype=torch.bool)
    mask[..., :1224] = True
    rope = helpers.rope_tables(1024, hd)
    p = RectifiedHunyuanVideoSpaAttnProcessor2_0("torch", 2, None, 0.3, 0)
    o, e = p(helpers.fake_attn(101, heads, hd, added=True), hs, enc, mask, rope)
    close(o, "hy_dual_out"); close(e, "hy_dual_enc")
    assert p.current_step == 1
    p = RectifiedHunyuanVideoSpaAttnProcessor2_0("vanilla", 2, None, 0.3, 25)
    o, e = p(helpers.fake_attn(102, heads, hd, added=False), hs, enc, mask, rope)
    close(o, "hy_single_out"); close(e, "hy_single_enc")


@torch.no_grad()
def test_flux_config1_shape_dense_cpu():
    """Flux.1-dev 512x512: 1024 image + 512 text tokens, dense CPU processor path (BASELINE configs[0])."""
    rope = helpers.rope_tables(1536, hd)
    p = RectifiedFluxSpaAttnProcessor2_0("torch", 2, None, 0.3, 0, 512)
    o, e = p(helpers.fake_attn(103, heads, hd, added=True), helpers.hidden(103, 20, 1, 1024, dim),
             helpers.hidden(103, 21, 1, 512, dim), None, rope)
    close(o, "fx_dual_out"); close(e, "fx_dual_enc")
    p = RectifiedFluxSpaAttnProcessor2_0("vanilla", 2, None, 0.3, 40, 512)
    o = p(helpers.fake_attn(104, heads, hd, added=False), helpers.hidden(104, 22, 1, 1536, dim), None, None, rope)
    close(o, "fx_single_out")


@torch.no_grad()
def test_wan21_self_and_cross():
    a = helpers.fake_attn(105, heads, hd, wan=True)
    p = RectifiedWanT2VSpaAttnProcessor2_0("torch", 2, None, 0.3, 5, 1)
    close(p(a, helpers.hidden(105, 20, 1, 900, dim), None, None, helpers.wan_freqs(900, hd)), "wan_self_out")
    close(p(a, helpers.hidden(105, 20, 1, 900, dim), helpers.hidden(105, 23, 1, 512, dim), None, None),
          "wan_cross_out")
    assert p.current_step == 2
    with pytest.raises(ImportError):
        RectifiedWanT2VSpaAttnProcessor2_0("bogus", 2, None, 0.3)(a, helpers.hidden(105, 20, 1, 128, dim))


@torch.no_grad()
def test_cogvideo_dense_warmup():
    p = RectifiedCogVideoXVideoSpaAttnProcessor2_0("torch", 2, None, 0.3, 0)
    o, e = p(helpers.fake_attn(106, 4, 64, added=False), helpers.hidden(106, 20, 1, 768, 256),
             helpers.hidden(106, 21, 1, 226, 256), None, helpers.rope_tables(768, 64))
    close(o, "cog_out"); close(e, "cog_enc")


def test_step_counters_wrap_like_reference():
    from rectified_spaattn_amd import rectified_wan22_attn as w22
    p = RectifiedHunyuanVideoSpaAttnProcessor2_0("torch", 2, None, 0.3)
    p.current_step = 49
    a = helpers.fake_attn(1, 1, 64, added=False)
    with torch.no_grad():
        p(a, helpers.hidden(1, 1, 1, 128, 64), None, None, None)
    assert p.current_step == 0
    t = w22.RectifiedWanT2VSpaAttnProcessor2_0("sparse", 2, None, 0.3, 3, 0, warm_steps=4)
    assert not t._use_sparse()
    t.current_step = 4
    assert t._use_sparse()
    t.processor_id = 40
    assert not t._use_sparse()
    assert t._wrap == 80 and w22.RectifiedWanTI2VSpaAttnProcessor2_0("sparse", 2, None, 0.3)._wrap == 100
    f = RectifiedFluxSpaAttnProcessor2_0("sparse", 2, None, 0.3, 37)
    assert f.processor_id == 37 and f.text_length == 256


def test_get_set_attn_processors():
    class Att(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.proc = "old"

        def get_processor(self):
            return self.proc

        def set_processor(self, p):
            self.proc = p

    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.attn1, self.attn2, self.ff = Att(), Att(), torch.nn.Linear(2, 2)

    m = torch.nn.Module()
    m.blocks = torch.nn.ModuleList([Blk(), Blk()])
    got = attn_processor.get_attn_processors(m)
    assert sorted(got) == ["blocks.0.attn1.processor", "blocks.0.attn2.processor", "blocks.1.attn1.processor",
                           "blocks.1.attn2.processor"]
    attn_processor.set_attn_processor(m, "new")
    assert set(attn_processor.get_attn_processors(m).values()) == {"new"}
    attn_processor.set_attn_processor(m, {k: i for i, k in enumerate(sorted(got))})
    assert attn_processor.get_attn_processors(m)["blocks.1.attn2.processor"] == 3
    with pytest.raises(ValueError):
        attn_processor.set_attn_processor(m, {"blocks.0.attn1.processor": 1})


def test_fullattn_cpu_modes_match_reference_vectors():
    from rectified_spaattn_amd import attn, synth
    from rectified_spaattn_amd._lib import RsaError
    z = np.load(os.path.join(GOLDEN, "dense_1536.npz"))
    q, k, v = (torch.from_numpy(x) for x in synth.structured_qkv(int(z["seed"]), 1, 1, 1536, 128))
    am = torch.zeros(1, 1, 1, 1536, dtype=torch.bool)
    am[..., : int(z["n_valid"])] = True
    np.testing.assert_allclose(attn.fullattn(q, k, v, mode="torch").numpy(), z["torch"], atol=2e-5)
    np.testing.assert_allclose(attn.fullattn(q, k, v, mode="vanilla").numpy(), z["torch"], atol=2e-5)
    np.testing.assert_allclose(attn.fullattn(q, k, v, mode="vanilla", attn_mask=am).numpy(), z["vanilla_masked"],
                               atol=2e-5)
    np.testing.assert_allclose(attn.fullattn(q, k, v, mode="torch", attn_mask=am).numpy(), z["vanilla_masked"],
                               atol=2e-5)
    np.testing.assert_allclose(attn.fullattn(q, k, v, mode="torch", causal=True).numpy(), z["torch_causal"], atol=2e-5)
    np.testing.assert_allclose(attn.fullattn(q, k, v, mode="vanilla", causal=True).numpy(), z["torch_causal"], atol=2e-5)
    with pytest.raises(NotImplementedError):
        attn.fullattn(q, k, v, mode="nope")
    with pytest.raises(RsaError):
        attn.fullattn(q, k, v, mode="flash")
    assert attn.get_cu_seqlens(100, 20, [5, 7], device="cpu").tolist() == [0, 105, 120, 227, 240]
    m = attn.get_attn_mask(4, 3, [1, 2], device="cpu")
    assert m.shape == (2, 1, 1, 7) and m[0, 0, 0].tolist() == [True] * 5 + [False] * 2
    assert attn.get_flash_attn_params(4, 3, [1], device="cpu")[2:] == (7, 7)


def test_cabi_library_exports_every_declared_symbol():
    """include/rsa.h <-> librsa_hip.so: every declared entry point is exported (no compute calls on CPU)."""
    import re
    from rectified_spaattn_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "rsa.h")).read()
    declared = set(re.findall(r"\b(rsa_[a-z_0-9]+)\s*\(", hdr)) - {"rsa_status"}
    assert declared == set(_lib.EXPORTED), declared ^ set(_lib.EXPORTED)
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rsa_version() >= 100
    assert lib.rsa_status_string(-2).decode().startswith("unsupported")
    # argument validation happens on the host, before any launch
    lay = _lib.RsaLayout(1, 1, 96, 128, 1, 1, 0, 128, 128, 1, 0, 0, 128, 0)
    import ctypes
    sizes = (ctypes.c_size_t * _lib.NUM_BUFFERS)()
    tot = ctypes.c_size_t()
    assert lib.rsa_buffer_bytes(ctypes.byref(lay), ctypes.byref(sizes), ctypes.byref(tot)) == -2
    lay.D = 128
    assert lib.rsa_buffer_bytes(ctypes.byref(lay), ctypes.byref(sizes), ctypes.byref(tot)) == 0 and tot.value > 0
    lay.NB_total = 7
    assert lib.rsa_buffer_bytes(ctypes.byref(lay), ctypes.byref(sizes), ctypes.byref(tot)) == -1


def test_cabi_fp8_entry_points_validate_on_the_host():
    """The fp8 entry points reject bad layouts / head dims / workspaces before any launch (runs without a GPU)."""
    import ctypes
    from rectified_spaattn_amd import _lib
    lib = _lib.lib()
    lay = _lib.RsaLayout(1, 2, 128, 256, 2, 2, 0, 256, 256, 2, 0, 0, 256, 0)
    s4, tot = (ctypes.c_size_t * 4)(), ctypes.c_size_t()
    assert lib.rsa_fp8_operand_bytes(ctypes.byref(lay), ctypes.byref(s4), ctypes.byref(tot)) == 0
    assert list(s4)[:3] == [2 * 256 * 128] * 3 and tot.value >= 3 * 2 * 256 * 128
    lay64 = _lib.RsaLayout(1, 2, 64, 256, 2, 2, 0, 256, 256, 2, 0, 0, 256, 0)
    assert lib.rsa_fp8_operand_bytes(ctypes.byref(lay64), ctypes.byref(s4), ctypes.byref(tot)) == 0    # head_dim 64 is served
    assert list(s4)[:3] == [2 * 256 * 64] * 3
    lay32 = _lib.RsaLayout(1, 2, 32, 256, 2, 2, 0, 256, 256, 2, 0, 0, 256, 0)
    assert lib.rsa_fp8_operand_bytes(ctypes.byref(lay32), ctypes.byref(s4), ctypes.byref(tot)) == -2   # head_dim 32 is not
    ops = _lib.RsaFp8Operands()
    assert lib.rsa_carve_fp8_operands(ctypes.byref(lay), None, 0, ctypes.byref(ops)) == -1             # null workspace
    assert lib.rsa_carve_fp8_operands(ctypes.byref(lay), ctypes.c_void_p(4096), 16, ctypes.byref(ops)) == -3  # too small
    t = _lib.RsaTensor4(0, 0, 0, 0)
    assert lib.rsa_quantize_fp8(ctypes.byref(lay), t, t, t, None, None) == -1                          # null operands
    assert lib.rsa_quantize_fp8(ctypes.byref(lay), t, t, t, ctypes.byref(ops), None) == -1             # null images
    d = ctypes.c_size_t()
    assert lib.rsa_dense_fp8_bytes(1, 2, 300, 500, 128, ctypes.byref(d)) == 0
    assert d.value >= 2 * (384 + 2 * 512) * 128
    assert lib.rsa_dense_fp8_bytes(1, 2, 300, 500, 64, ctypes.byref(d)) == 0 and d.value >= 3 * 2 * 512 * 64
    assert lib.rsa_dense_fp8_bytes(1, 2, 300, 500, 32, ctypes.byref(d)) == -2
    assert lib.rsa_dense_fp8_bytes(0, 2, 300, 500, 128, ctypes.byref(d)) == -1
    # the tuning hook is inert unless the process opted in with RSA_TUNING=1 (then unknown keys are bad arguments)
    assert lib.rsa_set_tuning(b"no_such_key", 1) == (-1 if os.environ.get("RSA_TUNING") == "1" else -2)


# ---- round 2: Wan2.2 processors and B = 2 against reference vectors (tests/golden/processors_r2.npz) ------------------
G2 = np.load(os.path.join(GOLDEN, "processors_r2.npz"))


def close2(a, name, atol=3e-3):
    np.testing.assert_allclose(a.detach().numpy(), G2[name].astype(np.float32), atol=atol, rtol=2e-3)


@torch.no_grad()
def test_wan22_processors_dense_cpu():
    """RectifiedWan{TI2V,T2V,I2V}SpaAttnProcessor2_0.__call__ (reference rectified_wan22_attn.py:29-163, :181-288,
    :306-413): projections through the non-fused helper, full-width RMSNorm, cos/sin RoPE on [B,S,H,D], dense CPU."""
    from rectified_spaattn_amd import rectified_wan22_attn as w22
    a = helpers.fake_attn(111, heads, hd, wan=True)
    hs = helpers.hidden(111, 20, 1, 900, dim)
    rope = helpers.wan22_rope(900, hd)
    for cls, args in ((w22.RectifiedWanTI2VSpaAttnProcessor2_0, (3, 1)), (w22.RectifiedWanT2VSpaAttnProcessor2_0, (5, 0, 2)),
                      (w22.RectifiedWanI2VSpaAttnProcessor2_0, (7, 2, 0))):
        p = cls("torch", 2, None, 0.3, *args)
        close2(p(a, hs, None, None, rope), "w22_ti2v_dense")      # dense attention: the same value for every class
        assert p.current_step == 1
    # cross attention (the reference ran it in "flash" mode = exact attention over the 512 text keys)
    p = w22.RectifiedWanT2VSpaAttnProcessor2_0("torch", 2, None, 0.3, 5, 0)
    close2(p(a, hs, helpers.hidden(111, 23, 1, 512, dim), None, None), "w22_cross")
    # warm-up layers / steps of a "sparse" processor are dense: same numbers as the dense vector
    assert np.abs(G2["w22_ti2v_warm"].astype(np.float32) - G2["w22_ti2v_dense"].astype(np.float32)).max() < 2e-3
    assert np.abs(G2["w22_t2v_layer40_dense"].astype(np.float32) - G2["w22_ti2v_dense"].astype(np.float32)).max() < 2e-3


@torch.no_grad()
def test_wan_image_context_branch_is_the_intended_attention():
    """Deliberate deviation: the reference's Wan2.2 image-context branch passes query as [B,S,H,D] to
    SDPA against [B,H,S_img,D] keys (rectified_wan22_attn.py:94-97) and raises unless S == H (recorded by
    make_golden.py: w22_imgctx_reference_runs == 0); Wan2.1's branch (rectified_wan21_attn.py:443-449) transposes first.
    This package uses the Wan2.1 form for both: out = attention(q, text keys) + attention(q, image keys)."""
    from rectified_spaattn_amd import rectified_wan22_attn as w22
    assert int(G2["w22_imgctx_reference_runs"]) == 0
    a = helpers.fake_attn_wan_i2v(112, heads, hd)
    hs, enc = helpers.hidden(111, 20, 1, 300, dim), helpers.hidden(112, 24, 1, 512 + 17, dim)
    p = w22.RectifiedWanI2VSpaAttnProcessor2_0("torch", 2, None, 0.3, 7, 0)
    got = p(a, hs, enc, None, None)      # cross attention: no RoPE
    # the same thing written out with plain torch ops
    q = a.norm_q(a.to_q(hs)).unflatten(2, (heads, -1))
    k = a.norm_k(a.to_k(enc[:, 17:])).unflatten(2, (heads, -1))
    v = a.to_v(enc[:, 17:]).unflatten(2, (heads, -1))
    ki = a.norm_added_k(a.add_k_proj(enc[:, :17])).unflatten(2, (heads, -1))
    vi = a.add_v_proj(enc[:, :17]).unflatten(2, (heads, -1))
    sd = torch.nn.functional.scaled_dot_product_attention
    o = sd(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)) + sd(q.transpose(1, 2), ki.transpose(1, 2),
                                                                          vi.transpose(1, 2))
    want = a.to_out[0](o.transpose(1, 2).flatten(2, 3))
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=2e-5)


@torch.no_grad()
def test_cogvideo_batch_of_two_dense():
    """CogVideoX under classifier-free guidance: B = 2, cu_seqlens [0, S, 2S] (reference :478-498)."""
    p = RectifiedCogVideoXVideoSpaAttnProcessor2_0("torch", 2, None, 0.3, 0)
    o, e = p(helpers.fake_attn(106, 4, 64, added=False), helpers.hidden(106, 30, 2, 768, 256),
             helpers.hidden(106, 31, 2, 226, 256), None, helpers.rope_tables(768, 64))
    close2(o, "cog_b2_out"); close2(e, "cog_b2_enc")


def test_layout_specs_refuse_geometries_the_reference_cannot_run():
    """Host logic only: the layouts the reference's own slicing cannot serve raise ValueError here instead of reaching the library
    (rectified_hunyuan_attn.py:313-332 with no valid text token; rectified_cogvideo_attn.py:318-320,:359-366 with visual tokens that
    do not fill whole blocks; the un-padded reshapes of the HunyuanVideo / Flux paths)."""
    from rectified_spaattn_amd import _core
    with pytest.raises(ValueError):
        _core.LayoutSpec.hunyuan(6 * 128 + 256, 6 * 128)          # 0 valid text tokens
    with pytest.raises(ValueError):
        _core.LayoutSpec.hunyuan(6 * 128 + 200, 6 * 128 + 100)    # S % 128 != 0
    with pytest.raises(ValueError):
        _core.LayoutSpec.flux(5 * 128 + 100, 128)
    with pytest.raises(ValueError):
        _core.LayoutSpec.cogvideo(7 * 128, 226)                   # 670 visual tokens: the text rows would start inside a visual block
    ok = _core.LayoutSpec.cogvideo(6 * 128 + 226, 226)            # the shipped shape: visual tokens a multiple of 128, 30 pad rows
    assert (ok.NB_total, ok.NBv, ok.q_text_valid) == (8, 6, 226)
    one = _core.LayoutSpec.cogvideo(128 + 226, 226)
    assert (one.NB_total, one.NBv) == (3, 1)
