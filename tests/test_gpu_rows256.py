"""The 256-row form of the 64-row K5 for DENSE calls (round 6; rsa_attn_kernel64.hip NW = 4, both head dims, tuning key k5_rows256): four waves, one
per SIMD, on ONE K/V ring -- every half-tile staged once per 256 query rows.  A row's arithmetic does not depend on the tile it sits
in (same 64-row waves, same key order, same reference), so for a plain dense call the two forms must agree BYTE FOR BYTE; with a
causal limit or two segments the tiles differ in which kept blocks take the boundary path, still the same arithmetic per row."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(flag, fn):
    from rectified_spaattn_amd import _lib
    L = _lib.lib()
    try:
        assert L.rsa_set_tuning(b"k5_rows256", flag) == 0
        out = fn().clone()
        torch.cuda.synchronize()
        return out
    finally:
        L.rsa_set_tuning(b"k5_rows256", 1)


def _ref(q, k, v, q_split=None, kv_split=None, causal=False):
    B, H, Sq, D = q.shape
    Sk = k.shape[2]
    q_split = Sq if q_split is None else q_split
    kv_split = Sk if kv_split is None else kv_split
    out = torch.zeros(B, Sq, H, D, dtype=torch.float64, device=q.device)
    for (r0, r1, c0, c1) in ((0, q_split, 0, kv_split), (q_split, Sq, kv_split, Sk)):
        if r1 <= r0:
            continue
        sc = q[:, :, r0:r1].double() @ k[:, :, c0:c1].double().transpose(2, 3) * float(D) ** -0.5
        if causal:
            i = torch.arange(r1 - r0, device=q.device)[:, None]
            j = torch.arange(c1 - c0, device=q.device)[None, :]
            sc = sc.masked_fill(j > i + ((c1 - c0) - (r1 - r0)), float("-inf"))
        w = torch.softmax(sc, dim=-1)
        w = torch.nan_to_num(w, nan=0.0)                     # a row that sees no key: zeros
        out[:, r0:r1] = (w @ v[:, :, c0:c1].double()).transpose(1, 2)
    return out


@pytest.mark.parametrize("Sq,Sk", [(257, 300), (512, 512), (1000, 777), (1536, 4096), (3001, 129)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D", [128, 64])
def test_plain_dense_calls_agree_byte_for_byte_with_the_128_row_form(Sq, Sk, dt, D):
    from rectified_spaattn_amd import _core
    g = torch.Generator(device=DEV).manual_seed(Sq + Sk)
    H = 3
    q = torch.randn(1, H, Sq, D, generator=g, device=DEV).to(dt)
    k, v = (torch.randn(1, H, Sk, D, generator=g, device=DEV).to(dt) for _ in range(2))
    a = _run(1, lambda: _core.dense_attention(q, k, v))
    b = _run(0, lambda: _core.dense_attention(q, k, v))
    assert torch.equal(a, b)
    err = (a.double() - _ref(q, k, v)).abs()
    tol = 2e-2 if dt == torch.bfloat16 else 2e-3
    assert float(err.max()) <= tol and float(err.mean()) <= tol / 10


@pytest.mark.parametrize("case", [dict(Sq=900, Sk=900, causal=True), dict(Sq=700, Sk=1100, causal=True),
                                  dict(Sq=1280, Sk=1280, q_split=1024, kv_split=1024), dict(Sq=1111, Sk=999, q_split=300, kv_split=640),
                                  dict(Sq=1111, Sk=999, q_split=300, kv_split=640, causal=True)])
@pytest.mark.parametrize("D", [128, 64])
def test_causal_and_two_segment_calls(case, D):
    from rectified_spaattn_amd import _core
    Sq, Sk = case["Sq"], case["Sk"]
    g = torch.Generator(device=DEV).manual_seed(Sq * 7 + Sk)
    H = 2
    q = torch.randn(1, H, Sq, D, generator=g, device=DEV).to(torch.bfloat16)
    k, v = (torch.randn(1, H, Sk, D, generator=g, device=DEV).to(torch.bfloat16) for _ in range(2))
    kw = dict(q_split=case.get("q_split"), kv_split=case.get("kv_split"), causal=case.get("causal", False))
    a = _run(1, lambda: _core.dense_attention(q, k, v, **kw))
    b = _run(0, lambda: _core.dense_attention(q, k, v, **kw))
    ref = _ref(q, k, v, **kw)
    for o in (a, b):
        err = (o.double() - ref).abs()
        assert float(err.max()) <= 2e-2 and float(err.mean()) <= 2e-3, (case, float(err.max()))
    assert float((a.float() - b.float()).abs().max()) <= 2e-2


def test_strided_heads_batch_and_the_public_entry_point():
    """[B, S, H, D]-strided views (what split_heads hands over), B = 2, through fullattn(mode='flash')."""
    from rectified_spaattn_amd import attn
    g = torch.Generator(device=DEV).manual_seed(4)
    B, H, S, D = 2, 4, 1300, 128
    q, k, v = (torch.randn(B, S, H, D, generator=g, device=DEV).to(torch.bfloat16).transpose(1, 2) for _ in range(3))
    a = _run(1, lambda: attn.fullattn(q, k, v, mode="flash"))
    b = _run(0, lambda: attn.fullattn(q, k, v, mode="flash"))
    assert torch.equal(a, b)
    ref = torch.cat([_ref(q[i:i + 1], k[i:i + 1], v[i:i + 1]) for i in range(B)]).transpose(1, 2)
    assert float((a.double() - ref).abs().max()) <= 2e-2
