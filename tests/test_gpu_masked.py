"""fullattn with an attn_mask that depends on the query row, or an additive float one (reference attn.py:101-106 hands torch's
SDPA any mask broadcastable to [b, a, s, s1]; :134-147 builds the same bias for "vanilla"): served on the device by
rsa_dense_masked_fwd.  Checked against torch's own SDPA in float32 on the CPU, on the 2-byte-rounded inputs (a floating-point
kernel: tolerance as for the other dense paths, 2e-2 on N(0,1) values)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _qkv(B, H, S, S1, D, dt, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, H, S, D, generator=g).to(dt)
    k = torch.randn(B, H, S1, D, generator=g).to(dt)
    v = torch.randn(B, H, S1, D, generator=g).to(dt)
    return q, k, v


def _ref(q, k, v, mask):
    m = mask if mask.dtype == torch.bool else mask.float()
    return F.scaled_dot_product_attention(q.float(), k.float(), v.float(), attn_mask=m)


@pytest.mark.parametrize("mode", ["torch", "vanilla"])
@pytest.mark.parametrize("D,dt", [(128, torch.bfloat16), (128, torch.float16), (64, torch.bfloat16)])
def test_boolean_mask_that_depends_on_the_query_row(mode, D, dt):
    from rectified_spaattn_amd import attn
    B, H, S, S1 = 2, 3, 200, 333          # ragged: neither a multiple of the 128-row workgroup nor of the 32-key tile
    q, k, v = _qkv(B, H, S, S1, D, dt, 1)
    g = torch.Generator().manual_seed(2)
    for shape in ((B, 1, S, S1), (1, H, S, S1), (B, H, S, S1), (1, 1, S, S1)):
        mask = torch.rand(shape, generator=g) < 0.6
        mask[..., 0] = True                                   # every row keeps a key
        out = attn.fullattn(q.to(DEV), k.to(DEV), v.to(DEV), mode=mode, attn_mask=mask.to(DEV))
        assert out.shape == (B, H, S, D) and out.dtype == dt
        err = (out.float().cpu() - _ref(q, k, v, mask)).abs()
        assert err.max() <= 2e-2, (shape, float(err.max()))


@pytest.mark.parametrize("mode", ["torch", "vanilla"])
def test_additive_float_mask(mode):
    from rectified_spaattn_amd import attn
    B, H, S, S1, D = 1, 4, 160, 257, 128
    q, k, v = _qkv(B, H, S, S1, D, torch.bfloat16, 3)
    g = torch.Generator().manual_seed(4)
    for shape, mdt in (((B, H, S, S1), torch.bfloat16), ((1, 1, S, S1), torch.float32), ((B, 1, 1, S1), torch.bfloat16)):
        bias = (2.0 * torch.randn(shape, generator=g)).to(mdt)
        bias[..., 5] = float("-inf")                          # -inf entries = masked keys
        out = attn.fullattn(q.to(DEV), k.to(DEV), v.to(DEV), mode=mode, attn_mask=bias.to(DEV))
        # (the reference casts a float mask to the dtype of q before SDPA, attn.py:102-103: so does the device path)
        err = (out.float().cpu() - _ref(q, k, v, bias.to(torch.bfloat16))).abs()
        assert err.max() <= 2e-2, (shape, float(err.max()))


def test_a_row_without_any_key():
    """torch's fused SDPA ("torch" mode) returns zeros for such a row (torch >= 2.5), the explicit softmax of "vanilla" NaN: the
    device path gives what the reference's own CPU expression of the mode gives here."""
    from rectified_spaattn_amd import attn
    q, k, v = _qkv(1, 2, 64, 96, 128, torch.bfloat16, 5)
    mask = torch.ones(1, 1, 64, 96, dtype=torch.bool)
    mask[0, 0, 7] = False
    mask[0, 0, 40, 32:] = False                               # attended keys only in the first tile
    keep = [i for i in range(64) if i != 7]
    for mode in ("torch", "vanilla"):
        ref = attn.fullattn(q.float(), k.float(), v.float(), mode=mode, attn_mask=mask)       # CPU tensors: the reference's expression
        out = attn.fullattn(q.to(DEV), k.to(DEV), v.to(DEV), mode=mode, attn_mask=mask.to(DEV)).float().cpu()
        assert torch.equal(torch.isnan(out[:, :, 7]), torch.isnan(ref[:, :, 7])), mode
        if not torch.isnan(ref[:, :, 7]).any():
            assert torch.equal(out[:, :, 7], ref[:, :, 7]), mode
        assert (out[:, :, keep] - ref[:, :, keep]).abs().max() <= 2e-2, mode


def test_c_abi_float32_masks_and_bad_arguments():
    """The C entry also takes float32 masks as they are (hosts other than this Python wrapper), and refuses what it cannot serve."""
    import ctypes
    from rectified_spaattn_amd import _core, _lib
    q, k, v = (x.to(DEV) for x in _qkv(1, 2, 100, 130, 128, torch.bfloat16, 6))
    bias = torch.randn(1, 1, 100, 130, generator=torch.Generator().manual_seed(7))
    out = _core.dense_attention_masked(q, k, v, bias.to(DEV))            # float32 kind
    ref = _ref(q.cpu(), k.cpu(), v.cpu(), bias).transpose(1, 2)
    assert (out.float().cpu() - ref).abs().max() <= 2e-2
    L = _lib.lib()
    o4 = _lib.RsaOut4(out.data_ptr(), out.stride(0), out.stride(2), out.stride(1))
    args = lambda D, kind, mp: (1, 2, 100, 130, D, _core.dtype_code(q.dtype), _core._t4(q), _core._t4(k), _core._t4(v), mp, kind,  # noqa: E731
                                0, 0, 130, 1, 1, o4, _core._stream())
    assert L.rsa_dense_masked_fwd(*args(128, 3, None)) == -1            # no mask
    assert L.rsa_dense_masked_fwd(*args(128, 9, bias.to(DEV).data_ptr())) == -1     # unknown kind
    assert L.rsa_dense_masked_fwd(*args(32, 3, bias.to(DEV).data_ptr())) == -2      # head dim not built


@pytest.mark.parametrize("S,S1", [(1, 1), (33, 31), (129, 32), (5, 700)])
def test_tiny_and_ragged_shapes(S, S1):
    from rectified_spaattn_amd import attn
    q, k, v = _qkv(1, 2, S, S1, 64, torch.float16, 8)
    g = torch.Generator().manual_seed(9)
    mask = torch.rand(1, 2, S, S1, generator=g) < 0.5
    mask[..., S1 - 1] = True
    out = attn.fullattn(q.to(DEV), k.to(DEV), v.to(DEV), mode="torch", attn_mask=mask.to(DEV))
    assert (out.float().cpu() - _ref(q, k, v, mask)).abs().max() <= 2e-2


# ---- dropout (round 5): fullattn(drop_rate=...) on the device, attn.py:104-106 / :148 -----------------------------------------
def test_dropout_zero_rate_and_full_rate_edges():
    """drop_rate = 0 through the dropout entry point equals the same kernel without dropout; drop_rate = 1 drops every weight
    (torch.dropout(p = 1) returns zeros)."""
    from rectified_spaattn_amd import _core
    q, k, v = (x.to(DEV) for x in _qkv(1, 2, 150, 201, 128, torch.bfloat16, 5))
    mask = (torch.rand(1, 1, 150, 201, generator=torch.Generator().manual_seed(6)) < 0.7)
    mask[..., 0] = True
    base = _core.dense_attention_masked(q, k, v, mask.to(DEV))
    assert torch.equal(_core.dense_attention_dropout(q, k, v, 0.0, 123, mask.to(DEV)), base)
    assert float(_core.dense_attention_dropout(q, k, v, 1.0, 123, mask.to(DEV)).abs().max()) == 0.0
    # no mask at all, causal: against torch on the CPU
    o = _core.dense_attention_dropout(q[:, :, :150], k[:, :, :150], v[:, :, :150], 0.0, 1, None, causal=True).transpose(1, 2)
    ref = F.scaled_dot_product_attention(q[:, :, :150].float().cpu(), k[:, :, :150].float().cpu(), v[:, :, :150].float().cpu(), is_causal=True)
    assert float((o.float().cpu() - ref).abs().max()) <= 2e-2


@pytest.mark.parametrize("mode", ["torch", "vanilla"])
def test_dropout_is_reproducible_unbiased_and_drops_the_stated_share(mode):
    """The distribution the reference's semantics fix (independent keeps with probability 1 - p, kept weights x 1 / (1 - p)):
    (i) same torch seed -> same bytes, another seed -> other bytes; (ii) with V = 1 every output row is sum of kept weights /
    (1 - p): its mean over rows and heads is 1 and the share of dropped weights, read off uniform attention, is p; (iii) the
    mean over 48 seeds approaches the dropout-free output."""
    from rectified_spaattn_amd import attn
    B, H, S, S1, D, p = 1, 4, 256, 512, 128, 0.3
    q, k, v = _qkv(B, H, S, S1, D, torch.bfloat16, 9)
    dq, dk, dv = q.to(DEV), k.to(DEV), v.to(DEV)
    torch.manual_seed(77)
    a = attn.fullattn(dq, dk, dv, mode=mode, drop_rate=p)
    torch.manual_seed(77)
    b = attn.fullattn(dq, dk, dv, mode=mode, drop_rate=p)
    c = attn.fullattn(dq, dk, dv, mode=mode, drop_rate=p)
    assert torch.equal(a, b) and not torch.equal(a, c)
    # uniform attention (q = 0), V = 1: out = (#kept / S1) / (1 - p)
    ones = torch.ones_like(dv)
    u = attn.fullattn(torch.zeros_like(dq), dk, ones, mode=mode, drop_rate=p).float()
    kept_share = u[..., 0] * (1 - p)                         # per (head, row): share of the 512 keys kept
    assert abs(float(kept_share.mean()) - (1 - p)) < 0.01
    assert abs(float(kept_share.std()) - (p * (1 - p) / S1) ** 0.5) < 0.006      # binomial spread: keeps are independent
    assert float((u - u[..., :1]).abs().max()) == 0.0
    # unbiased: the mean over seeds tends to the dropout-free result
    base = attn.fullattn(dq, dk, dv, mode=mode).float()
    acc = torch.zeros_like(base)
    n = 48
    for _ in range(n):
        acc += attn.fullattn(dq, dk, dv, mode=mode, drop_rate=p).float()
    err = (acc / n - base).abs()
    assert float(err.mean()) < 0.03 and float(err.max()) < 0.25, (float(err.mean()), float(err.max()))


def test_dropout_with_masks_and_causal():
    """A boolean row mask and causal attention under dropout: dropped weights never resurrect masked keys (V = one-hot over the
    keys shows which keys contribute), rows keep summing to ~1 on average."""
    from rectified_spaattn_amd import attn
    B, H, S, D = 1, 2, 128, 128
    q, k, _ = _qkv(B, H, S, S, D, torch.bfloat16, 11)
    v = torch.zeros(B, H, S, D, dtype=torch.bfloat16)
    v[..., torch.arange(S), torch.arange(S) % D] = 1.0          # key j writes into channel j (S == D here)
    out = attn.fullattn(q.to(DEV), k.to(DEV), v.to(DEV), mode="torch", drop_rate=0.25, causal=True).float().cpu()
    upper = torch.triu(torch.ones(S, S, dtype=torch.bool), diagonal=1)
    assert float(out[0, 0][upper].abs().max()) == 0.0, "a key above the diagonal contributed"
    mask = torch.rand(B, 1, S, S, generator=torch.Generator().manual_seed(3)) < 0.5
    mask[..., 0] = True
    out = attn.fullattn(q.to(DEV), k.to(DEV), v.to(DEV), mode="vanilla", drop_rate=0.25, attn_mask=mask.to(DEV)).float().cpu()
    assert float(out[0, 1][~mask[0, 0]].abs().max()) == 0.0, "a masked key contributed"
    assert abs(float(out.sum(-1).mean()) - 1.0) < 0.05


def test_a_mask_on_another_device_is_refused_with_the_librarys_error_type():
    """this branch raised NameError (RsaError was not imported in _core.py).  Both entry points -- the masked
    one and its dropout twin -- must refuse a CPU mask for device tensors with RsaError, and fullattn must pass it through."""
    from rectified_spaattn_amd import _core, _lib, attn
    g = torch.Generator(device="cuda:0").manual_seed(3)
    q, k, v = (torch.randn(1, 2, 256, 128, generator=g, device="cuda:0").to(torch.bfloat16) for _ in range(3))
    cpu_mask = torch.ones(1, 1, 256, 256, dtype=torch.bool)
    with pytest.raises(_lib.RsaError, match="device of q"):
        _core.dense_attention_masked(q, k, v, cpu_mask)
    with pytest.raises(_lib.RsaError, match="device of q"):
        _core.dense_attention_dropout(q, k, v, 0.1, 7, mask=cpu_mask)
    with pytest.raises(_lib.RsaError):
        attn.fullattn(q, k, v, mode="torch", attn_mask=cpu_mask)
