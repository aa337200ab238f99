"""Producers / consumers either side of the attention path on the HIP library:

  permute_tokens     hidden_states[:, order]  (Hilbert permute in / out; scripts/main_hunyuan.py:88, :183)
  build_attention_mask   the [B,1,1,N] key-padding mask + valid lengths (scripts/main_hunyuan.py:91-103)
  qk_norm_rope       per-head RMSNorm + rotary embedding (+ placement into the [visual | text] concat) in one pass
                     (rectified_hunyuan_attn.py:452-498, rectified_flux_attn.py:443-484)
"""
import ctypes
from typing import Optional, Tuple

import torch

from . import _core, _lib
from ._lib import RsaOut4


def _order_i32(order: torch.Tensor, device) -> torch.Tensor:
    """int32 device copy of an index vector, cached on the source tensor (scripts keep it as a long CUDA tensor)."""
    key = (order._version, str(device))
    cache = getattr(order, "_rsa_i32", None)
    if cache is None or cache[0] != key:
        cache = (key, order.to(device=device, dtype=torch.int32).contiguous())
        try:
            order._rsa_i32 = cache
        except AttributeError:
            pass
    return cache[1]


def permute_tokens(x: torch.Tensor, order: torch.Tensor) -> torch.Tensor:
    """x [B, S, C] -> x[:, order]  ([B, len(order), C]) with one HBM pass on the HIP gather kernel."""
    _core._require_device(x)
    _core.dtype_code(x.dtype)
    B, S, C = x.shape
    if x.stride(-1) != 1 or C % 8 or x.stride(0) % 8 or x.stride(1) % 8 or x.data_ptr() % 16:
        x = x.contiguous()
    if C % 8:
        raise AssertionError("permute_tokens needs a row length that is a multiple of 8 elements")
    idx = _order_i32(order, x.device)
    n = idx.numel()
    out = torch.empty((B, n, C), dtype=x.dtype, device=x.device)
    vp = ctypes.c_void_p
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().rsa_permute_tokens(B, n, C, vp(x.data_ptr()), x.stride(0), x.stride(1),
                                                 vp(idx.data_ptr()), vp(out.data_ptr()), out.stride(0), out.stride(1),
                                                 _core._stream()), "rsa_permute_tokens")
    return out


def build_attention_mask(latent_len: int, encoder_attention_mask: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """([B,1,1,N] bool mask, [B] valid lengths): latent tokens always valid, text tokens up to each sample's count."""
    B, n_txt = encoder_attention_mask.shape
    eff = latent_len + encoder_attention_mask.sum(dim=1, dtype=torch.int)
    idx = torch.arange(latent_len + n_txt, device=encoder_attention_mask.device)[None, :]
    return (idx < eff[:, None])[:, None, None, :], eff


def norm_params(norm) -> Optional[Tuple[Optional[torch.Tensor], float]]:
    """(weight | None, eps) of an RMSNorm-like module, or None if the module is something else (then the caller
    keeps the module call)."""
    if norm is None:
        return None
    if type(norm).__name__ not in ("RMSNorm", "RMS") or not hasattr(norm, "eps"):
        return None
    if getattr(norm, "bias", None) is not None:
        return None
    if norm.eps is None:   # torch.nn.RMSNorm(eps=None) resolves eps per input dtype: keep the module call
        return None
    w = getattr(norm, "weight", None)
    if w is not None and w.dtype == torch.float32:
        # diffusers multiplies by an fp32 weight WITHOUT first rounding the normalised value to the activation dtype;
        # the fused kernel reproduces the 2-byte-weight order of operations only
        return None
    return w, float(norm.eps)


def layernorm_params(norm):
    """(weight | None, bias | None, eps) of a torch.nn.LayerNorm-like module over the last dim, or None."""
    if norm is None or type(norm).__name__ not in ("LayerNorm", "FP32LayerNorm") or not hasattr(norm, "eps"):
        return None
    shape = tuple(getattr(norm, "normalized_shape", ()))
    if len(shape) != 1:
        return None
    w, b = getattr(norm, "weight", None), getattr(norm, "bias", None)
    if type(norm).__name__ == "FP32LayerNorm":
        return None   # computes on an fp32 copy and rounds differently: keep the module call
    return w, b, float(norm.eps)


def qk_norm_rope(x: torch.Tensor, heads: int, norm=None, rotary=None, rope_tokens: Optional[int] = None,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [B, S, H*D] projection -> [B, H, S, D] view of a [B, S, H, D] buffer holding RMSNorm(x) rotated.

    norm: (weight [D] | None, eps) for RMSNorm, (weight | None, bias | None, eps) for LayerNorm (layernorm_params), or
    None (no normalisation); rotary: (cos, sin) fp32 [>= rope_tokens, D] or None;
    rope_tokens: tokens [0, rope_tokens) are rotated (default all); out: optional [B, S, H, D] destination (e.g. a
    slice of the concat buffer), else a new one."""
    _core._require_device(x)
    B, S, HD = x.shape
    D = HD // heads
    xv = x.view(B, S, heads, D).transpose(1, 2)  # [B,H,S,D] view
    if xv.stride(-1) != 1 or any(s % 8 for s in xv.stride()[:-1]) or xv.data_ptr() % 16:
        xv = xv.contiguous()
    if out is None:
        out = torch.empty((B, S, heads, D), dtype=x.dtype, device=x.device)
    assert out.shape == (B, S, heads, D) and out.dtype == x.dtype and out.stride(-1) == 1
    w = bias = eps = None
    layer_norm = norm is not None and len(norm) == 3      # (weight, bias, eps) from layernorm_params
    if norm is not None:
        if layer_norm:
            w, bias, eps = norm
            if bias is not None:
                bias = bias.detach().to(device=x.device, dtype=torch.float32).contiguous()
        else:
            w, eps = norm
        if w is not None:
            w = w.detach().to(device=x.device, dtype=torch.float32).contiguous()
    cos = sin = None
    if rotary is not None:
        cos, sin = rotary
        cos = cos.to(device=x.device, dtype=torch.float32).contiguous()
        sin = sin.to(device=x.device, dtype=torch.float32).contiguous()
        if rope_tokens is None:
            rope_tokens = S
        assert cos.shape[-1] == D and cos.shape[0] >= rope_tokens
    vp = ctypes.c_void_p
    o4 = RsaOut4(out.data_ptr(), out.stride(0), out.stride(2), out.stride(1))
    if layer_norm:
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().rsa_qk_layernorm_rope(
                B, heads, S, D, _core.dtype_code(x.dtype), _core._t4(xv), vp(w.data_ptr()) if w is not None else None,
                vp(bias.data_ptr()) if bias is not None else None, ctypes.c_float(eps),
                vp(cos.data_ptr()) if cos is not None else None, vp(sin.data_ptr()) if sin is not None else None,
                int(rope_tokens or 0), o4, _core._stream()), "rsa_qk_layernorm_rope")
        return out.transpose(1, 2)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().rsa_qk_norm_rope(
            B, heads, S, D, _core.dtype_code(x.dtype), _core._t4(xv), vp(w.data_ptr()) if w is not None else None,
            ctypes.c_float(eps if eps is not None else 0.0), 1 if norm is not None else 0,
            vp(cos.data_ptr()) if cos is not None else None, vp(sin.data_ptr()) if sin is not None else None,
            int(rope_tokens or 0), o4, _core._stream()), "rsa_qk_norm_rope")
    return out.transpose(1, 2)


def norm_rope_across_heads(x: torch.Tensor, heads: int, norm=None, rotary=None) -> torch.Tensor:
    """The Wan producers in one pass (rsa_norm_rope_heads): x [B, S, H*D] projection -> [B, H, S, D] view of a new
    [B, S, H, D] buffer holding RMSNorm-across-heads(x) rotated per head.

    norm: (weight [H*D] | None, eps) from norm_params, or None; rotary: None, a complex128 tensor [..., S, D/2] (Wan2.1's
    `rotary_emb`, any leading singleton dims) or a (cos, sin) pair of fp32 tensors [..., S, ..., D] with per-pair values
    duplicated (Wan2.2's `rotary_emb`)."""
    _core._require_device(x)
    B, S, HD = x.shape
    D = HD // heads
    if x.stride(-1) != 1 or x.stride(0) % 8 or x.stride(1) % 8 or x.data_ptr() % 16:
        x = x.contiguous()
    out = torch.empty((B, S, heads, D), dtype=x.dtype, device=x.device)
    w = eps = None
    if norm is not None:
        w, eps = norm
        if w is not None:
            w = w.detach().to(device=x.device, dtype=torch.float32).contiguous()
            assert w.numel() == HD
    kind, fa, fb = 0, None, None
    if rotary is not None:
        if torch.is_tensor(rotary):
            assert rotary.is_complex() and rotary.dtype == torch.complex128 and rotary.numel() == S * (D // 2)
            fa = torch.view_as_real(rotary.to(x.device).reshape(S, D // 2)).contiguous()   # [S, D/2, 2] doubles
            kind = 1
        else:
            cos, sin = rotary
            assert cos.numel() == S * D and sin.numel() == S * D
            fa = cos.to(device=x.device, dtype=torch.float32).reshape(S, D).contiguous()
            fb = sin.to(device=x.device, dtype=torch.float32).reshape(S, D).contiguous()
            kind = 2
    vp = ctypes.c_void_p
    o4 = RsaOut4(out.data_ptr(), out.stride(0), out.stride(2), out.stride(1))
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().rsa_norm_rope_heads(
            B, heads, S, D, _core.dtype_code(x.dtype), vp(x.data_ptr()), x.stride(0), x.stride(1),
            vp(w.data_ptr()) if w is not None else None, ctypes.c_float(eps if eps is not None else 0.0),
            1 if norm is not None else 0, kind, vp(fa.data_ptr()) if fa is not None else None,
            vp(fb.data_ptr()) if fb is not None else None, o4, _core._stream()), "rsa_norm_rope_heads")
    return out.transpose(1, 2)
