"""Dense attention entry point `fullattn` -- same signature, layouts and modes as the reference's
rectified_spaattn/attn.py:60-154, served by the HIP dense kernel (rsa_dense_fwd) for device tensors.

Modes on DEVICE tensors all run the same gfx950 kernel (there is no flash-attn / SDPA dependency):
  "flash"   two-segment varlen semantics from cu_seqlens_q / cu_seqlens_kv (attn.py:107-120)
  "torch"   plain attention, optional boolean key-padding mask [b,1,1,s1] (attn.py:101-106)
  "vanilla" same result as "torch" (attn.py:121-149)
CPU tensors: "torch" and "vanilla" keep the reference's own CPU behaviour (plain PyTorch ops: this is
BASELINE config 1, the CPU-runnable plumbing case); "flash" needs the device and raises.
"""
import math

import torch
import torch.nn.functional as F

from . import _core, _operator
from ._lib import RsaError

# pre/post layout adapters kept for API parity (attn.py:18-31).  "flash" packs [b,a,s,d] -> [(b s), a, d].
MEMORY_LAYOUT = {
    "flash": (lambda x: x.transpose(1, 2).reshape(x.shape[0] * x.shape[2], x.shape[1], x.shape[3]),
              lambda x: x.transpose(1, 2)),
    "torch": (lambda x: x, lambda x: x),
    "vanilla": (lambda x: x, lambda x: x),
}


def get_cu_seqlens(img_seq_len, txt_seq_len, text_len, device="cuda"):
    """[0, s_0, L, L + s_1, 2L, ...] with s_i = img_seq_len + text_len[i], L = img_seq_len + txt_seq_len
    (attn.py:34-57)."""
    batch_size = len(text_len)
    max_len = img_seq_len + txt_seq_len
    vals = [0]
    for i in range(batch_size):
        vals.append(i * max_len + int(text_len[i]) + img_seq_len)
        vals.append((i + 1) * max_len)
    return torch.tensor(vals, dtype=torch.int32, device=device)


def get_attn_mask(img_seq_len, txt_seq_len, text_len, device="cuda"):
    """bool [B,1,1,N], True for image tokens and the first text_len[i] text tokens (attn.py:156-163)."""
    n = img_seq_len + txt_seq_len
    lens = torch.tensor([img_seq_len + int(t) for t in text_len], device=device)
    return (torch.arange(n, device=device)[None, :] < lens[:, None])[:, None, None, :]


def get_flash_attn_params(img_seq_len, txt_seq_len, text_len, device="cuda"):
    """(cu_seqlens_q, cu_seqlens_kv, max_seqlen_q, max_seqlen_kv) (attn.py:165-171)."""
    cu = get_cu_seqlens(img_seq_len, txt_seq_len, text_len, device=device)
    n = img_seq_len + txt_seq_len
    return cu, cu, n, n


def _to_list(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        return [int(i) for i in x.tolist()]  # device tensor -> one host sync; pass ints/lists to avoid it
    return [int(i) for i in x]


def _device_dense(q, k, v, splits):
    """q [b,a,s,d], k/v [b,a,s1,d]; splits: per batch item (q_split, kv_split).  Returns [b,a,s,d] view."""
    B = q.shape[0]
    fp8 = _operator.DENSE_FP8 and q.shape[-1] == 128  # set_dense_fp8(): e4m3 operands on the fp8 MFMA
    if len(set(splits)) == 1:
        return _core.dense_attention(q, k, v, splits[0][0], splits[0][1], qkv_fp8=fp8).transpose(1, 2)
    outs = [_core.dense_attention(q[i:i + 1], k[i:i + 1], v[i:i + 1], *splits[i], qkv_fp8=fp8) for i in range(B)]
    return torch.cat(outs, 0).transpose(1, 2)


def _key_padding_counts(attn_mask, B, S1):
    """bool mask broadcastable to [b,a,s,s1] that only depends on the key index and is a prefix mask
    (what get_attn_mask builds) -> valid key count per batch item; anything else is not supported."""
    m = attn_mask
    if m.dtype != torch.bool or m.dim() != 4 or m.shape[1] != 1 or m.shape[2] != 1 or m.shape[3] != S1:
        raise NotImplementedError("device fullattn supports only boolean key-padding masks [b,1,1,s1]")
    m = m.reshape(m.shape[0], S1)
    counts = m.sum(-1)
    prefix = (m == (torch.arange(S1, device=m.device)[None, :] < counts[:, None])).all()
    counts = [int(c) for c in counts.tolist()]  # host sync, as the reference's .item() (hunyuan :502)
    if not bool(prefix):
        raise NotImplementedError("device fullattn supports only prefix (padding) key masks")
    if len(counts) == 1 and B > 1:
        counts = counts * B
    return counts


def fullattn(q, k, v, mode="flash", drop_rate=0, attn_mask=None, causal=False, cu_seqlens_q=None,
             cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None, batch_size=1):
    """QKV attention.  q [b,a,s,d], k/v [b,a,s1,d] -> [b,a,s,d]  (same contract as attn.py:60-154)."""
    if mode not in MEMORY_LAYOUT:
        raise NotImplementedError(f"Unsupported attention mode: {mode}")
    B, _, S, D = q.shape
    S1 = k.shape[2]
    if q.is_cuda:
        if drop_rate:
            raise NotImplementedError("dropout is not implemented in the HIP attention path")
        if causal:
            raise NotImplementedError("causal attention is not implemented in the HIP attention path")
        if mode == "flash":
            cq, ck = _to_list(cu_seqlens_q), _to_list(cu_seqlens_kv)
            if cq is None or ck is None:
                splits = [(S, S1)] * B
            else:  # segment pair i: [cu[2i], cu[2i+1]) and [cu[2i+1], cu[2i+2]) inside batch item i
                splits = [(cq[2 * i + 1] - i * S, ck[2 * i + 1] - i * S1) for i in range(B)]
        else:
            if attn_mask is None:
                splits = [(S, S1)] * B
            else:
                splits = [(S, c) for c in _key_padding_counts(attn_mask, B, S1)]
        return _device_dense(q, k, v, splits)
    # ---- CPU tensors: the reference's CPU-runnable modes ----
    if mode == "flash":
        raise RsaError("fullattn(mode='flash') needs device tensors (HIP kernel); use mode='torch' on CPU")
    if mode == "torch":
        if attn_mask is not None and attn_mask.dtype != torch.bool:
            attn_mask = attn_mask.to(q.dtype)
        return F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, dropout_p=drop_rate, is_causal=causal)
    # vanilla: explicit softmax(q k^T / sqrt(d) + bias) v
    bias = torch.zeros(B, q.shape[1], S, S1, dtype=q.dtype, device=q.device)
    if causal:
        assert attn_mask is None, "Causal mask and attn_mask cannot be used together"
        bias.masked_fill_(~torch.ones(S, S, dtype=torch.bool, device=q.device).tril(), float("-inf"))
    if attn_mask is not None:
        if attn_mask.dtype == torch.bool:
            bias.masked_fill_(~attn_mask, float("-inf"))
        else:
            bias += attn_mask
    w = ((q @ k.transpose(-2, -1)) / math.sqrt(D) + bias).softmax(dim=-1)
    w = torch.dropout(w, p=drop_rate, train=True)
    return w @ v
