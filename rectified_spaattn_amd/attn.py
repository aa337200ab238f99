"""Dense attention entry point `fullattn` -- same signature, layouts and modes as the reference's
rectified_spaattn/attn.py:60-154, served by the HIP dense kernel (rsa_dense_fwd) for device tensors.

causal=True is served for "torch" / "vanilla" (per-row key limits of the same kernel).  In mode "flash" it is IGNORED, as in the
reference: attn.py:107-116 hands flash_attn_varlen_func only the cu_seqlens / max_seqlen arguments, so its flash mode is never
causal (a warning says so once).
Modes on DEVICE tensors all run the same gfx950 kernel (there is no flash-attn / SDPA dependency):
  "flash"   two-segment varlen semantics from cu_seqlens_q / cu_seqlens_kv (attn.py:107-120)
  "torch"   plain attention, optional boolean key-padding mask [b,1,1,s1] (attn.py:101-106)
  "vanilla" same result as "torch" (attn.py:121-149)
CPU tensors: "torch" and "vanilla" keep the reference's own CPU behaviour (plain PyTorch ops: this is
BASELINE config 1, the CPU-runnable plumbing case); "flash" needs the device and raises.
"""
import math
import warnings

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
_WARNED_FLASH_CAUSAL = False


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


def _splits_from_cu(cq, ck, B, S, S1):
    """cu_seqlens_q / cu_seqlens_kv are segment boundaries over the packed [(b s), a, d] rows that
    flash_attn_varlen_func receives (attn.py:107-120): segment j = q rows [cq[j], cq[j+1]) attending kv rows
    [ck[j], ck[j+1]).  The HIP dense kernel serves, per batch item, one or two such segments: rows < q_split attend
    kv [0, kv_split), rows >= q_split attend kv [kv_split, S1).  Accepted forms (all the reference builds):
    get_cu_seqlens' 2B+1 entries (:34-57), the processors' 3 entries -- [0, valid, S] for B = 1 (hunyuan :503) and
    [0, S, S*B] (cogvideo :481, flux, wan) where every batch item is one full segment -- and B+1 entries."""
    if len(cq) != len(ck):
        raise NotImplementedError(f"cu_seqlens_q / cu_seqlens_kv of different lengths ({len(cq)}, {len(ck)})")
    segs = [(cq[j], cq[j + 1], ck[j], ck[j + 1]) for j in range(len(cq) - 1) if cq[j + 1] > cq[j]]
    splits = []
    for i in range(B):
        q0, q1, k0, k1 = i * S, (i + 1) * S, i * S1, (i + 1) * S1
        mine = [sg for sg in segs if sg[0] < q1 and sg[1] > q0]
        ok = bool(mine) and mine[0][0] == q0 and mine[-1][1] == q1 and len(mine) <= 2
        if ok and len(mine) == 1:
            a, b, c, d = mine[0]
            ok = c == k0 and k0 <= d <= k1
            splits.append((S, d - k0))
        elif ok:
            (a, b, c, d), (a2, b2, c2, d2) = mine
            ok = b == a2 and c == k0 and d == c2 and d2 == k1 and k0 <= d <= k1
            splits.append((b - q0, d - k0))
        if not ok:
            raise NotImplementedError(
                f"fullattn(mode='flash') on device: cu_seqlens_q={cq}, cu_seqlens_kv={ck} do not describe one or two "
                f"segments per batch item (B={B}, S={S}, S1={S1}); supported: get_cu_seqlens' 2B+1 entries, "
                f"[0, valid, S] for B = 1, [0, S, S*B], or B+1 entries")
    return splits


def _device_dense(q, k, v, splits, dense_fp8=None, causal=False):
    """q [b,a,s,d], k/v [b,a,s1,d]; splits: per batch item (q_split, kv_split).  Returns [b,a,s,d] view."""
    B = q.shape[0]
    # e4m3 operands on the fp8 MFMA: per call, else the process default of set_dense_fp8()
    fp8 = _operator._fp8_mode(_operator.DENSE_FP8 if dense_fp8 is None else dense_fp8, q.shape[-1])
    if len(set(splits)) == 1:
        return _core.dense_attention(q, k, v, splits[0][0], splits[0][1], qkv_fp8=fp8, causal=causal).transpose(1, 2)
    outs = [_core.dense_attention(q[i:i + 1], k[i:i + 1], v[i:i + 1], *splits[i], qkv_fp8=fp8, causal=causal)
            for i in range(B)]
    return torch.cat(outs, 0).transpose(1, 2)


def _key_mask_rows(attn_mask, B, S1):
    """bool mask broadcastable to [b,a,s,s1] that only depends on the key index ([b,1,1,s1], what get_attn_mask builds)
    -> ([b, s1] bool rows, valid key count per batch item, whether every row is a prefix); None for any other mask."""
    m = attn_mask
    if m.dtype != torch.bool or m.dim() != 4 or m.shape[1] != 1 or m.shape[2] != 1 or m.shape[3] != S1:
        return None          # not a key mask: the general-mask kernel serves it (_core.dense_attention_masked)
    m = m.reshape(m.shape[0], S1)
    counts = m.sum(-1)
    prefix = (m == (torch.arange(S1, device=m.device)[None, :] < counts[:, None])).all()
    host = torch.cat([counts, prefix.reshape(1).to(counts.dtype)]).tolist()  # ONE host sync, as the reference's .item() (hunyuan :502)
    counts, prefix = [int(c) for c in host[:-1]], bool(host[-1])
    if len(counts) == 1 and B > 1:
        counts = counts * B
        m = m.expand(B, S1)
    return m, counts, prefix


def _compact_keys(k, v, rows, counts):
    """Key masks with holes: the valid keys of every batch item moved to the front (a softmax does not see the order of its
    keys), so the kernel's prefix limit serves them.  No host synchronisation: a stable argsort of the inverted mask lists the
    valid key indices first, in order; one gather per tensor; the rows behind an item's count are other (finite) keys that lie
    beyond its limit."""
    n = max(counts)
    order = torch.argsort((~rows).to(torch.uint8), dim=1, stable=True)[:, :n]          # [B, n]
    idx = order[:, None, :, None].expand(-1, k.shape[1], -1, k.shape[3])
    return torch.gather(k, 2, idx), torch.gather(v, 2, idx)


def fullattn(q, k, v, mode="flash", drop_rate=0, attn_mask=None, causal=False, cu_seqlens_q=None,
             cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None, batch_size=1, dense_fp8=None):
    """QKV attention.  q [b,a,s,d], k/v [b,a,s1,d] -> [b,a,s,d]  (same contract as attn.py:60-154).
    dense_fp8 (not in the reference): per-call choice of e4m3 operands for the device kernel; None = the process
    default set by set_dense_fp8()."""
    if mode not in MEMORY_LAYOUT:
        raise NotImplementedError(f"Unsupported attention mode: {mode}")
    B, _, S, D = q.shape
    S1 = k.shape[2]
    if q.is_cuda:
        if drop_rate and mode != "flash":
            # attn.py:104-106 (SDPA's dropout_p) / :148 (torch.dropout on the softmax's output): the plain device kernel with a
            # counter-based keep mask, seeded from torch's generator -- reproducible under torch.manual_seed, not torch's own stream
            if not 0.0 <= float(drop_rate) <= 1.0:
                raise ValueError(f"dropout probability has to be between 0 and 1, but got {drop_rate}")
            if causal and attn_mask is not None:
                if mode == "vanilla":
                    raise AssertionError("Causal mask and attn_mask cannot be used together")      # attn.py:127-129
                raise NotImplementedError("device fullattn: causal together with an attn_mask (torch's SDPA refuses the combination too)")
            if causal and S != S1 and mode == "vanilla":
                raise NotImplementedError("device fullattn: vanilla's causal triangle is s x s (attn.py:130): needs s == s1")
            m = attn_mask
            if m is not None and m.dtype != torch.bool:
                m = m.to(q.dtype)
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
            return _core.dense_attention_dropout(q, k, v, float(drop_rate), seed, m, causal=bool(causal),
                                                 empty_rows_nan=(mode == "vanilla")).transpose(1, 2)
        # (mode "flash" does not forward drop_rate: attn.py:107-116 calls flash_attn_varlen_func without it)
        if causal and mode != "flash" and (attn_mask is not None or S != S1):
            # "torch" / "vanilla" put the causal triangle top-left (attn.py:105, :129-133; vanilla asserts no mask); the
            # kernel's segments are bottom-right aligned like flash-attn: the two agree only for s == s1 without padding
            raise NotImplementedError("device fullattn: causal with mode 'torch' / 'vanilla' needs s == s1 and no attn_mask")
        if mode == "flash":
            cq, ck = _to_list(cu_seqlens_q), _to_list(cu_seqlens_kv)
            if cq is None or ck is None:
                splits = [(S, S1)] * B
            else:
                splits = _splits_from_cu(cq, ck, B, S, S1)
        else:
            if attn_mask is None:
                splits = [(S, S1)] * B
            else:
                km = _key_mask_rows(attn_mask, B, S1)
                if km is None:
                    # a mask that depends on the query row, or an additive one (attn.py:101-106, :134-147): the plain kernel
                    if causal:
                        raise NotImplementedError("device fullattn: causal together with a row-dependent attn_mask "
                                                  "(torch's SDPA refuses the combination too)")
                    if attn_mask.dtype != torch.bool:
                        attn_mask = attn_mask.to(q.dtype)      # as the reference does before SDPA (attn.py:102-103)
                    # a row without attended keys: NaN from "vanilla"'s explicit softmax (attn.py:148), zeros from the fused
                    # SDPA of "torch" mode (torch >= 2.5)
                    return _core.dense_attention_masked(q, k, v, attn_mask, empty_rows_nan=(mode == "vanilla")).transpose(1, 2)
                rows, counts, prefix = km
                if min(counts) == 0:
                    # (the reference's SDPA returns NaN for a row without keys: nothing a caller can use -- refuse clearly)
                    raise ValueError("fullattn: attn_mask leaves a batch item without any key")
                if not prefix:
                    k, v = _compact_keys(k, v, rows, counts)
                splits = [(S, c) for c in counts]
        if causal and mode == "flash":
            # the reference's flash branch does not forward `causal` (attn.py:107-116): same call, same (non-causal) result
            global _WARNED_FLASH_CAUSAL
            if not _WARNED_FLASH_CAUSAL:
                _WARNED_FLASH_CAUSAL = True
                warnings.warn("fullattn(mode='flash', causal=True): the reference ignores `causal` in flash mode "
                              "(attn.py:107-116); so does this implementation -- use mode='torch' for causal attention")
            causal = False
        return _device_dense(q, k, v, splits, dense_fp8, causal=bool(causal))
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
