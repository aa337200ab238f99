"""Round-4 processor coverage on the device: the Flux processor's SPARSE branch for a sparse layer id on
both sides of the warm-up gate (`processor_id < 37 or >= 57`, reference rectified_flux_attn.py:493) and the dense warm-up
layer in between, and the Wan2.1 I2V processor with an image context (reference rectified_wan21_attn.py:512-632).

Checkers as in test_gpu_processors_r2.py: the oracle on the q / k / v the processor itself handed to the operator (2e-2 /
2e-3), the reference's kept mask of the same processor call first and its operator output on the query blocks whose kept set
agrees (tests/golden/processors_r4.npz, made by tests/golden/make_golden.py processors_r4 from the reference's own code)."""
import os

import numpy as np
import pytest
import torch

import helpers
from conftest import GOLDEN
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G4 = np.load(os.path.join(GOLDEN, "processors_r4.npz"))
heads, hd = 2, 128
dim = heads * hd


def _flux_inputs():
    a = helpers.attn_to(helpers.fake_attn(103, heads, hd, added=True), DEV, torch.bfloat16)
    hs = helpers.hidden(103, 20, 1, 1024, dim).to(DEV, torch.bfloat16)
    enc = helpers.hidden(103, 21, 1, 512, dim).to(DEV, torch.bfloat16)
    rope = tuple(t.to(DEV) for t in helpers.rope_tables(1536, hd))
    return a, hs, enc, rope


@torch.no_grad()
@pytest.mark.parametrize("pid", [0, 57])
def test_flux_sparse_processor_on_device(pid):
    from rectified_spaattn_amd import _core
    from rectified_spaattn_amd import rectified_flux_attn as fx
    from rectified_spaattn_amd import synth
    from test_gpu_processors_r2 import _capture, _device_mask
    assert int(G4["fx_sparse_id57_equals_id0"]) == 1     # the reference gives ids 0 and 57 the same answer on the same inputs
    a, hs, enc, rope = _flux_inputs()
    nbr = torch.from_numpy(synth.banded_neighbors(8, 1))
    p = fx.RectifiedFluxSpaAttnProcessor2_0("sparse", 2, nbr, 0.3, pid, 512)
    captured, restore = _capture(fx)
    try:
        o, e = p(a, hs, enc, None, rope)
    finally:
        restore()
    assert p.current_step == 1 and "qkv" in captured, "the sparse branch did not run"
    q, k, v = captured["qkv"]
    lay = orc.layout_flux(1536, 512)
    ref = orc.rectified_attention(q, k, v, lay, 2, 0.3, nbr.numpy())
    op_out = captured["out"].float().cpu().numpy()
    err = np.abs(op_out - ref)
    assert err.max() <= 2e-2 and err.mean() <= 2e-3, (err.max(), err.mean())
    assert o.shape == (1, 1024, dim) and e.shape == (1, 512, dim)
    full = torch.cat([o, e], 1).float().cpu().numpy()
    gold = np.concatenate([G4["fx_sparse_id0_out"], G4["fx_sparse_id0_enc"]], 1).astype(np.float32)
    assert np.abs(full - gold).mean() <= 2e-2      # vs the reference processor's own output (fp32 CPU run, may keep other blocks)
    # mask first, then the operator output on the rows whose kept set equals the reference's
    spec = _core.LayoutSpec.flux(1536, 512)
    kept = _device_mask(q, k, v, spec, 2, 0.3, nbr)[:, : spec.NBv]
    shape = tuple(int(x) for x in G4["fx_sparse_id0_mask_shape"])
    ref_kept = np.unpackbits(G4["fx_sparse_id0_mask"], axis=-1)[..., : shape[-1]].astype(bool)[0]
    assert ref_kept.shape == kept.shape, (ref_kept.shape, kept.shape)
    ref_out = G4["fx_sparse_id0_op_out"].astype(np.float32).reshape(1, 1536, heads, hd)
    got = op_out.reshape(1, 1536, heads, hd)
    flipped, compared, worst = 0, 0, 0.0
    for h in range(heads):
        for i in range(kept.shape[1]):
            if not np.array_equal(kept[h, i], ref_kept[h, i]):
                flipped += 1
                continue
            worst = max(worst, float(np.abs(got[0, i * 128:(i + 1) * 128, h] - ref_out[0, i * 128:(i + 1) * 128, h]).max()))
            compared += 1
    print(f"flux sparse id {pid}: {flipped} of {flipped + compared} (head, query block) rows keep another block set than the "
          f"reference; max |O - O_ref| on the agreeing rows {worst:.3e}")
    assert flipped <= 0.25 * (flipped + compared) and compared > 0 and worst <= 3e-2


@torch.no_grad()
def test_flux_warmup_layer_stays_dense_on_device():
    from rectified_spaattn_amd import rectified_flux_attn as fx
    from rectified_spaattn_amd import synth
    from test_gpu_processors_r2 import _capture
    a, hs, enc, rope = _flux_inputs()
    nbr = torch.from_numpy(synth.banded_neighbors(8, 1))
    p = fx.RectifiedFluxSpaAttnProcessor2_0("sparse", 2, nbr, 0.3, 40, 512)     # layers 37..56: dense
    captured, restore = _capture(fx)
    try:
        o, e = p(a, hs, enc, None, rope)
    finally:
        restore()
    assert "qkv" not in captured, "layer 40 must not take the sparse operator"
    assert np.abs(o.float().cpu().numpy() - G4["fx_sparse_id40_out"].astype(np.float32)).max() <= 6e-2
    assert np.abs(e.float().cpu().numpy() - G4["fx_sparse_id40_enc"].astype(np.float32)).max() <= 6e-2


@torch.no_grad()
def test_wan21_i2v_image_context_on_device():
    """Cross attention of a Wan2.1 I2V block: out = attention(q, 512 text keys) + attention(q, 257 image keys) -> to_out."""
    from rectified_spaattn_amd import rectified_wan21_attn as w21
    a = helpers.attn_to(helpers.fake_attn_wan_i2v(112, heads, hd), DEV, torch.bfloat16)
    hs = helpers.hidden(111, 20, 1, 900, dim).to(DEV, torch.bfloat16)
    enc = helpers.hidden(112, 24, 1, 257 + 512, dim).to(DEV, torch.bfloat16)
    p = w21.RectifiedWanI2VSpaAttnProcessor2_0("flash", 2, None, 0.3, 5, 0)
    o = p(a, hs, enc, None, None)
    assert o.shape == (1, 900, dim) and p.current_step == 1
    assert np.abs(o.float().cpu().numpy() - G4["wan21_i2v_imgctx"].astype(np.float32)).max() <= 6e-2
