"""Optimistic static softmax reference of the 64-row K5 (round 6; rsa_attn_kernel64.hip, tuning key k5_static): in bf16 the
steady-state loop keeps the reference m it is entered with and computes no row maxima; l and O are checked after the walk and a
workgroup whose walk overflowed walks again through the online body.  Same softmax either way:

  * ordinary data: k5_static on / off agree within rounding and each agrees with the oracle;
  * adversarial data -- scores that exceed the first keys' by far more than 127 binary orders, so exp2(S - m) IS infinite in the
    static body -- must come out exactly as well as through the online body: this passes only if the overflow is detected and
    the second pass runs (without it the rows are NaN);
  * fp16 never takes the static body (its P overflows at 2^16).
"""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _with_static(flag, fn):
    from rectified_spaattn_amd import _lib
    L = _lib.lib()
    try:
        assert L.rsa_set_tuning(b"k5_static", flag) == 0
        out = fn()
        torch.cuda.synchronize()
        return out
    finally:
        L.rsa_set_tuning(b"k5_static", 1)


def _dense_ref(q, k, v, kv_valid=None):
    qf, kf, vf = q[0].double(), k[0].double(), v[0].double()
    sc = qf @ kf.transpose(1, 2) * float(q.shape[-1]) ** -0.5
    if kv_valid is not None:
        sc[:, :, kv_valid:] = float("-inf")
    return (torch.softmax(sc, dim=-1) @ vf).transpose(0, 1)[None]          # [1, S, H, D]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_dense_static_and_online_agree_on_ordinary_data(dt):
    from rectified_spaattn_amd import _core
    g = torch.Generator(device=DEV).manual_seed(5)
    H, S, D = 3, 2048 + 77, 128
    q, k, v = (torch.randn(1, H, S, D, generator=g, device=DEV).to(dt) for _ in range(3))
    on = _with_static(1, lambda: _core.dense_attention(q, k, v).clone())
    off = _with_static(0, lambda: _core.dense_attention(q, k, v).clone())
    ref = _dense_ref(q, k, v)
    tol = 2e-2 if dt == torch.bfloat16 else 2e-3
    for o in (on, off):
        err = (o.double() - ref).abs()
        assert float(err.max()) <= tol and float(err.mean()) <= tol / 10
    if dt == torch.float16:
        assert torch.equal(on, off), "fp16 must not take the static body"


@pytest.mark.parametrize("where", ["late_keys", "one_row", "rising"])
def test_overflow_of_the_static_body_is_detected_and_redone(where):
    """late_keys: every key from 256 on scores ~200 binary orders above the first 32 keys (for every row); one_row: only one query
    row of one block sees such keys (the other 127 rows of its workgroup are redone with it); rising: the scores climb by ~60 binary
    orders per 128-key block (every block overflows a reference more than two blocks old)."""
    from rectified_spaattn_amd import _core
    g = torch.Generator(device=DEV).manual_seed(9)
    H, S, D = 2, 1536, 128
    u = torch.nn.functional.normalize(torch.randn(D, generator=g, device=DEV), dim=0)
    q = 0.3 * torch.randn(1, H, S, D, generator=g, device=DEV)
    k = 0.3 * torch.randn(1, H, S, D, generator=g, device=DEV)
    v = torch.randn(1, H, S, D, generator=g, device=DEV)
    if where == "late_keys":
        q = q + 8.0 * u
        k[:, :, 256:] += 200.0 * u
    elif where == "one_row":
        q[:, 1, 700] += 8.0 * u
        k[:, 1, 900:] += 200.0 * u
    else:
        q = q + 8.0 * u
        k = k + (torch.arange(S, device=DEV) // 128)[None, None, :, None] * 60.0 * u
    q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    ref = _dense_ref(q, k, v)
    on = _with_static(1, lambda: _core.dense_attention(q, k, v).clone())
    off = _with_static(0, lambda: _core.dense_attention(q, k, v).clone())
    assert torch.isfinite(on).all() and torch.isfinite(off).all()
    for o in (on, off):
        err = (o.double() - ref).abs()
        assert float(err.max()) <= 3e-2 and float(err.mean()) <= 3e-3, (where, float(err.max()), float(err.mean()))
    # a workgroup that was redone ran the online body: its rows are the online kernel's, byte for byte
    if where != "one_row":
        assert torch.equal(on, off)
    else:
        blk = 700 // 128
        assert torch.equal(on[0, blk * 128:(blk + 1) * 128, 1], off[0, blk * 128:(blk + 1) * 128, 1])


@pytest.mark.parametrize("layout", ["hunyuan", "wan"])
def test_sparse_operator_static_and_online_against_the_oracle(layout):
    """The rectified operator (sparse walks, text rows, the fused epilogue) with the static body on / off, both against the oracle;
    and the same call with one head's late keys blown up (the static walks of that head overflow and are redone)."""
    from bench import gen_qkv
    from rectified_spaattn_amd import _core, synth
    H, D, top_k, p = 2, 128, 6, 0.3
    S = 40 * 128 + (256 if layout == "hunyuan" else 53)
    q, k, v = gen_qkv(H, 0, S, S, D, torch.device(DEV), seed=17)
    if layout == "hunyuan":
        spec, lay = _core.LayoutSpec.hunyuan(S, S - 56), orc.layout_hunyuan(S, S - 56)
    else:
        spec, lay = _core.LayoutSpec.wan(S, 2), orc.layout_wan(S, 2)
    nbr = synth.banded_neighbors(spec.NBv, 2)
    for blow in (False, True):
        if blow:
            k = k.clone()
            k[:, 1, 20 * 128:] *= 24.0           # scores of head 1 against the later keys: tens to hundreds of binary orders up
        outs = [_with_static(f, lambda: _core.rectified_attention(q, k, v, spec, top_k, p, torch.from_numpy(nbr)).clone()) for f in (1, 0)]
        assert all(torch.isfinite(o).all() for o in outs)
        d = (outs[0].float() - outs[1].float()).abs()
        assert float(d.max()) <= 2e-2 and float(d.mean()) <= 2e-3, (layout, blow, float(d.max()), float(d.mean()))
        if blow:
            # (scores of hundreds of binary orders: the 2-byte rounding of the scaled q alone moves them by tenths -- every 2-byte
            # kernel, the reference's included, is then far from the fp64 oracle; what is checked is that the two bodies agree)
            continue
        ref = orc.rectified_attention(*(x.float().cpu().numpy() for x in (q, k, v)), lay, top_k, p, nbr)      # [1, S, H*D]
        for o in outs:
            err = np.abs(o.view(1, S, H * D).float().cpu().numpy() - ref)
            assert err.max() <= 2e-2 and err.mean() <= 2e-3, (layout, blow, err.max(), err.mean())


def test_redone_workgroups_inside_the_headline_launch():
    """The second pass at the SIZE of the headline launch: 24 heads x 902 blocks (aligned starts, text rows split, 43 generations),
    with the later keys of three heads blown up so that thousands of workgroups overflow their static walk and walk again while the
    others do not.  Static on / off must agree (same lists: the selection pass reads the same inputs) and stay finite; heads that
    were not touched are byte-identical between the two runs only where no workgroup was redone, so they are compared within
    rounding as well."""
    import bench
    from rectified_spaattn_amd import _core
    wl = bench.WORKLOADS["hunyuan_720p_128f"]
    spec = bench.make_spec(wl)
    q, k, v = bench.gen_inputs(wl, 24, 0, torch.device(DEV), "iid")
    for h in (3, 11, 22):
        k[:, h, 40000:] *= 16.0
    call = _core.StagedCall(q, k, v, spec, 90, 0.0, None, reuse_buffers=False)
    call.select()
    outs = []
    for f in (1, 0):
        outs.append(_with_static(f, lambda: call.attend().clone()))
    assert torch.isfinite(outs[0]).all() and torch.isfinite(outs[1]).all()
    d = (outs[0].float() - outs[1].float()).abs()
    # (blown-up keys make the softmax of those heads one-hot: outputs are single V rows of magnitude up to ~4, where two bf16 ulps are 0.03)
    ulp2 = 2 * 2.0 ** -7 * max(1.0, float(outs[1].float().abs().max()))
    assert float(d.max()) <= ulp2 and float(d.mean()) <= 4e-4, (float(d.max()), float(d.mean()), ulp2)
    chk = bench.check_output(call, spec)          # (the output of the last attend: the online body)
    assert chk["finite"]
