"""Shared implementation behind the four `rectified_block_sparse_attention` variants.

The reference keeps four near-identical copies of block_sparse_attention_combined (hunyuan :283-389,
flux :282-376, cogvideo :282-378, wan21 :276-357); here each variant only builds a LayoutSpec and calls the
one HIP pipeline (_core.rectified_attention)."""
from typing import Optional

import torch

from . import _core
from ._lib import BLOCK


def _int_at(x, idx: int, default: Optional[int] = None) -> int:
    """cu_seqlens may be a python sequence (no sync) or a tensor (one .item(), as in the reference)."""
    if x is None:
        if default is None:
            raise ValueError("cu_seqlens is required for this layout (reference appendix B-2)")
        return default
    v = x[idx]
    return int(v.item()) if isinstance(v, torch.Tensor) else int(v)


def _check_blocks(bm: int, bn: int):
    if bm != BLOCK or bn != BLOCK:
        raise NotImplementedError("the HIP path is built for block_size_M = block_size_N = 128 "
                                  "(the only value the reference scripts use)")


# PROCESS DEFAULT of the K5 operand precision (a processor / call can override it: processor.qkv_fp8, qkv_fp8=...):
# False = the input dtype (bf16/fp16, the
# reference's behaviour), True = e4m3 images of Q, K, V on the fp8 MFMA (head_dim 64 / 128; other head dims keep the
# 2-byte kernel).  Set with rectified_spaattn_amd.set_qkv_fp8(); the reference has no such switch (fp8 is its TODO).
QKV_FP8 = False


def set_qkv_fp8(enabled):
    """Switch the sparse operator (and therefore every processor's sparse steps) to fp8 K5 operands: True = e4m3 Q, K, V and P;
    "pv" = Q . K^T on the 2-byte inputs, e4m3 only for P . V (relative L1 0.04 of the layer output instead of 0.12, at 0.8 of the
    2-byte kernel's matrix work; head dims 64 and 128 -- the head dims 16 / 32 keep the 2-byte kernel, `_fp8_mode`); False = off.  Returns the previous setting."""
    global QKV_FP8
    if isinstance(enabled, str) and enabled != "pv":
        raise ValueError(f"set_qkv_fp8: False, True or 'pv', got {enabled!r}")
    old, QKV_FP8 = QKV_FP8, (enabled if isinstance(enabled, str) else bool(enabled))
    return old


# Same switch for the dense kernel behind fullattn(mode="flash" | device "torch"/"vanilla") -- separate because the dense
# (warm-up) steps are the ones a pipeline keeps exact on purpose.
DENSE_FP8 = False


def set_dense_fp8(enabled) -> bool:
    """False | True (e4m3 q, k, v, P) | "pv" (2-byte Q . K^T, e4m3 P . V)."""
    global DENSE_FP8
    if isinstance(enabled, str) and enabled != "pv":
        raise ValueError(f"set_dense_fp8: False, True or 'pv', got {enabled!r}")
    old, DENSE_FP8 = DENSE_FP8, (enabled if isinstance(enabled, str) else bool(enabled))
    return old


def _fp8_mode(choice, D: int):
    """False | True | "pv" for head dim D (head dims without the chosen kernel keep the 2-byte one)."""
    if isinstance(choice, str):
        if choice != "pv":
            raise ValueError(f"qkv_fp8: False, True or 'pv', got {choice!r}")
        return "pv" if D in (64, 128) else False
    return bool(choice) and D in (64, 128)


def run(variant: str, query, key, value, top_k, prob_threshold, block_neighbor_list, shape_xfuse,
        cu_seqlens_q=None, cu_seqlens_kv=None, text_length: int = 256, first_frame_blocks=None,
        block_size_M: int = 128, block_size_N: int = 128, qkv_fp8: Optional[bool] = None):
    """qkv_fp8: per-call choice of the K5 operand precision (None = the process default set_qkv_fp8())."""
    _check_blocks(block_size_M, block_size_N)
    B, H, S, D = query.shape
    if variant == "hunyuan":
        spec = _core.LayoutSpec.hunyuan(S, _int_at(cu_seqlens_q, 1))
    elif variant == "flux":
        spec = _core.LayoutSpec.flux(S, int(text_length), _int_at(cu_seqlens_kv, 1, S))
    elif variant == "cogvideo":
        spec = _core.LayoutSpec.cogvideo(S, int(text_length), _int_at(cu_seqlens_kv, 1, S))
    elif variant == "wan":
        spec = _core.LayoutSpec.wan(S, first_frame_blocks)
    else:
        raise ValueError(variant)
    return _core.rectified_attention(query, key, value, spec, int(top_k), float(prob_threshold),
                                     block_neighbor_list, shape_xfuse=shape_xfuse,
                                     qkv_fp8=_fp8_mode(QKV_FP8 if qkv_fp8 is None else qkv_fp8, D))


# ---- small helpers shared by the processors ---------------------------------------------------------
def split_heads(x: torch.Tensor, heads: int) -> torch.Tensor:
    """[B, S, H*D] -> [B, H, S, D] view (no copy; the kernels take the strides as they are)."""
    return x.unflatten(2, (heads, -1)).transpose(1, 2)


def rotary(x: torch.Tensor, freqs):
    """diffusers.models.embeddings.apply_rotary_emb when diffusers is installed, else the same maths for the
    (cos, sin) / use_real=True / unbind_dim=-1 convention the Hunyuan, Flux and CogVideoX pipelines use."""
    try:
        from diffusers.models.embeddings import apply_rotary_emb
        return apply_rotary_emb(x, freqs)
    except ImportError:
        cos, sin = freqs
        cos, sin = cos[None, None].to(x.device), sin[None, None].to(x.device)
        re, im = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
        rot = torch.stack([-im, re], dim=-1).flatten(3)
        return (x.float() * cos + rot.float() * sin).to(x.dtype)


_VALID_KEYS_MEMO = [None, -1, 0]  # weakref to the last mask tensor, its _version, its count


def valid_keys(attention_mask, default: int, num_true: Optional[int] = None) -> int:
    """attention_mask.sum().item() -- the host sync the reference pays in every layer (hunyuan :502).  Here it is paid
    once per forward: the transformer hands the SAME mask tensor object to all of its blocks
    (scripts/main_hunyuan.py:95-103, :128-150), so the count is remembered for that object (identity through a weak
    reference + the tensor's in-place version counter).  A caller that already knows the count passes num_true."""
    if num_true is not None:
        return int(num_true)
    if attention_mask is None:
        return default
    try:   # (inference-mode tensors have no version counter: reading it raises RuntimeError -- no memo for those)
        version = None if attention_mask.is_inference() else attention_mask._version
    except (RuntimeError, AttributeError):
        version = None
    if version is None:
        return int(attention_mask.sum().item())
    ref, ver, cnt = _VALID_KEYS_MEMO
    if ref is not None and ref() is attention_mask and ver == version:
        return cnt
    cnt = int(attention_mask.sum().item())
    import weakref
    try:
        _VALID_KEYS_MEMO[:] = [weakref.ref(attention_mask), version, cnt]   # one slot, replaced as a whole (GIL-atomic)
    except TypeError:
        _VALID_KEYS_MEMO[:] = [None, -1, 0]
    return cnt


# ---- fused producer path --------------------------------------------------------------
FUSED_PRODUCER = True  # processors use the fused RMSNorm + RoPE + concat kernel when its preconditions hold


def fused_qk_ok(x: torch.Tensor, heads: int, norms, rotary) -> bool:
    """Preconditions of glue.qk_norm_rope for a processor: device bf16/fp16 projections, head_dim 64/128, every
    norm either absent, RMSNorm-like or a LayerNorm over the head dim, rotary absent or a (cos, sin) pair of real tables."""
    from . import glue
    if not (FUSED_PRODUCER and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16)):
        return False
    if x.shape[-1] % heads or (x.shape[-1] // heads) not in (64, 128):
        return False
    for n in norms:
        if n is not None and glue.norm_params(n) is None and glue.layernorm_params(n) is None:
            return False
    if rotary is not None:
        if not (isinstance(rotary, (tuple, list)) and len(rotary) == 2 and all(torch.is_tensor(t) for t in rotary)):
            return False
        if rotary[0].dim() != 2 or rotary[0].is_complex():
            return False
    return True


def fused_heads_ok(x: torch.Tensor, heads: int, norms, rotary) -> bool:
    """Preconditions of glue.norm_rope_across_heads for the Wan processors: device bf16/fp16 projections, every norm
    absent or RMSNorm-like over the full inner dim with a 2-byte (or no) weight, rotary absent, one complex128 table
    (Wan2.1) or a pair of fp32 (cos, sin) tables (Wan2.2) of exactly S tokens."""
    from . import glue
    if not (FUSED_PRODUCER and x.is_cuda and x.dim() == 3 and x.dtype in (torch.bfloat16, torch.float16)):
        return False
    HD, S = x.shape[-1], x.shape[1]
    if HD % heads or (HD // heads) % 8 or 512 % (HD // heads) or HD > 8192:
        return False
    D = HD // heads
    for n in norms:
        if n is None:
            continue
        p = glue.norm_params(n)
        if p is None or (p[0] is not None and p[0].numel() != HD):
            return False
    if rotary is None:
        return True
    if torch.is_tensor(rotary):
        return rotary.is_complex() and rotary.dtype == torch.complex128 and rotary.numel() == S * (D // 2) \
            and rotary.shape[-1] == D // 2
    if isinstance(rotary, (tuple, list)) and len(rotary) == 2 and all(torch.is_tensor(t) for t in rotary):
        return all(t.dtype == torch.float32 and t.numel() == S * D and t.shape[-1] == D for t in rotary)
    return False


def norm_args(norm):
    """The `norm` argument of glue.qk_norm_rope for a processor's norm module (RMSNorm-like or LayerNorm), or None."""
    from . import glue
    if norm is None:
        return None
    p = glue.norm_params(norm)
    return p if p is not None else glue.layernorm_params(norm)
