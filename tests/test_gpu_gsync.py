"""Aligned starts of the sparse walks (rsa_attn.h, tuning key k5_gsync; round 4).  The workgroups of an XCD wait for their
generation before they stage their first tile, so that their ascending walks meet in the XCD's L2.  It is a scheduling aid:
the arithmetic and its order are untouched, so every kernel must give the same BYTES with it on and off -- in the 64-row kernel
(where it is on by default), in the 32-row kernel and in the e4m3 kernel (bit 1 of the key)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(H, nb, D, seed, dt):
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    S = nb * 128
    cent = torch.randn(H, nb, D, generator=g, device="cuda:0")

    def mk():
        return (cent.repeat_interleave(128, 1) + 0.7 * torch.randn(H, S, D, generator=g, device="cuda:0")).to(dt).view(1, H, S, D)
    return mk(), mk(), torch.randn(1, H, S, D, generator=g, device="cuda:0").to(dt)


@pytest.mark.parametrize("w64,D,dt", [(1, 128, torch.bfloat16), (1, 128, torch.float16), (0, 128, torch.bfloat16), (0, 64, torch.bfloat16),
                                       (3, 64, torch.bfloat16), (3, 64, torch.float16)])
def test_aligned_starts_do_not_change_a_byte(w64, D, dt):
    from rectified_spaattn_amd import _core, _lib
    H, nb, top_k = 8, 168, 14          # 1 344 workgroups (more than two generations of 8 x 64), 8 % of the keys kept: the walks wait
    q, k, v = _inputs(H, nb, D, 11, dt)
    spec = _core.LayoutSpec.wan(nb * 128, 0)
    L = _lib.lib()
    outs = []
    try:
        assert L.rsa_set_tuning(b"k5_w64", w64) == 0
        for gs in (0, 3, 0, 3):
            assert L.rsa_set_tuning(b"k5_gsync", gs) == 0
            out = _core.rectified_attention(q, k, v, spec, top_k, 0.05, None)
            torch.cuda.synchronize()
            outs.append(out.view(torch.int16).cpu().numpy().copy())
    finally:
        L.rsa_set_tuning(b"k5_w64", 3)
        L.rsa_set_tuning(b"k5_gsync", 1)
    assert np.isfinite(out.float().cpu().numpy()).all()
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


@pytest.mark.parametrize("mode", [True, "pv"], ids=["e4m3", "pv"])
def test_aligned_starts_do_not_change_a_byte_of_the_e4m3_path(mode):
    from rectified_spaattn_amd import _core, _lib
    H, nb, top_k = 8, 168, 14
    q, k, v = _inputs(H, nb, 128, 12, torch.bfloat16)
    spec = _core.LayoutSpec.wan(nb * 128, 0)
    L = _lib.lib()
    outs = []
    try:
        for gs in (0, 3, 0, 3):
            assert L.rsa_set_tuning(b"k5_gsync", gs) == 0
            out = _core.rectified_attention(q, k, v, spec, top_k, 0.05, None, qkv_fp8=mode)
            torch.cuda.synchronize()
            outs.append(out.view(torch.int16).cpu().numpy().copy())
    finally:
        L.rsa_set_tuning(b"k5_gsync", 1)
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


def test_a_walk_that_keeps_many_keys_and_a_small_grid_are_left_alone():
    """More than a fifth of the keys kept, or two generations or fewer: no counters are taken / nobody waits; same bytes again."""
    from rectified_spaattn_amd import _core, _lib
    L = _lib.lib()
    for H, nb, top_k in ((8, 168, 60), (2, 64, 6)):
        q, k, v = _inputs(H, nb, 128, 13, torch.bfloat16)
        spec = _core.LayoutSpec.wan(nb * 128, 0)
        outs = []
        try:
            for gs in (0, 1):
                assert L.rsa_set_tuning(b"k5_gsync", gs) == 0
                out = _core.rectified_attention(q, k, v, spec, top_k, 0.05, None)
                torch.cuda.synchronize()
                outs.append(out.view(torch.int16).cpu().numpy().copy())
        finally:
            L.rsa_set_tuning(b"k5_gsync", 1)
        assert np.array_equal(outs[0], outs[1])


def test_aligned_starts_inside_a_captured_graph():
    """The counters of a launch are cleared by a memset in stream order in front of it: captured, that is a memset node, and
    every replay starts from cleared counters.  Capture (no warm-up call of this shape first), replay twice on new inputs."""
    from rectified_spaattn_amd import _core
    H, nb, top_k = 8, 168, 14
    q, k, v = _inputs(H, nb, 128, 21, torch.bfloat16)
    q2, k2, v2 = _inputs(H, nb, 128, 22, torch.bfloat16)
    spec = _core.LayoutSpec.wan(nb * 128, 0)
    call = _core.StagedCall(q, k, v, spec, top_k, 0.05, None)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        call.select()                    # (torch's capture needs the allocator warm; K5 itself is NOT run before the capture)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        call.select()
        call.attend()
    for _ in range(2):
        q.copy_(q2); k.copy_(k2); v.copy_(v2)
        g.replay()
        torch.cuda.synchronize()
        eager = _core.rectified_attention(q2, k2, v2, spec, top_k, 0.05, None)
        assert torch.equal(call.out.reshape(eager.shape), eager)


def test_text_rows_first_or_last_and_aligned_starts_on_a_layout_with_text():
    """HunyuanVideo-like layout (visual blocks + a 256-token text tail, 200 valid): the split text-row pieces are the last
    workgroups of the grid (tuning key k5_text_last; first = the order of rounds 1-3), the visual walks wait for their
    generation: the work mapping and the waits change, no byte of the result does.  One head against the oracle as well."""
    from oracle import oracle as orc
    from rectified_spaattn_amd import _core, _lib
    H, nbv, top_k = 6, 190, 16
    S = (nbv + 2) * 128
    num_true = S - 56
    g = torch.Generator(device="cuda:0").manual_seed(31)
    cent = torch.randn(H, nbv + 2, 128, generator=g, device="cuda:0")

    def mk():
        return (cent.repeat_interleave(128, 1) + 0.7 * torch.randn(H, S, 128, generator=g, device="cuda:0")).to(torch.bfloat16).view(1, H, S, 128)
    q, k, v = mk(), mk(), torch.randn(1, H, S, 128, generator=g, device="cuda:0").to(torch.bfloat16)
    spec = _core.LayoutSpec.hunyuan(S, num_true)
    L = _lib.lib()
    outs = []
    try:
        # (the tail split -- tests/test_gpu_tail_split.py -- needs the text pieces last and changes the split blocks' rounding:
        # off here, where the claim is byte identity across the scheduling switches)
        assert L.rsa_set_tuning(b"k5_tail_split", 0) == 0
        for gs, tl in ((0, 0), (1, 1), (1, 0), (0, 1)):
            assert L.rsa_set_tuning(b"k5_gsync", gs) == 0 and L.rsa_set_tuning(b"k5_text_last", tl) == 0
            out = _core.rectified_attention(q, k, v, spec, top_k, 0.05, None, shape_xfuse=True)
            torch.cuda.synchronize()
            outs.append(out.view(torch.int16).cpu().numpy().copy())
    finally:
        L.rsa_set_tuning(b"k5_gsync", 1)
        L.rsa_set_tuning(b"k5_text_last", 1)
        L.rsa_set_tuning(b"k5_tail_split", 1)
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])
    out = _core.rectified_attention(q, k, v, spec, top_k, 0.05, None, shape_xfuse=True)    # the product's defaults (tail split on)
    lay = orc.layout_hunyuan(S, num_true)
    hd = 3
    qf, kf, vf = (x[0, hd].float().cpu().numpy() for x in (q, k, v))
    ref = orc.rectified_attention(qf[None, None], kf[None, None], vf[None, None], lay, top_k, 0.05, None)
    got = out[0, :, hd].float().cpu().numpy()
    err = np.abs(got - ref.reshape(got.shape))
    assert err.max() <= 2e-2 and err.mean() <= 1.5e-3
