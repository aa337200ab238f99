"""CPU restatement of the reference's dense path, used as the bench's cpu_baseline (kind "port").

TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/oracle.py header).  The reference's `fullattn(mode="torch")`
(attn.py:101-106) is a direct call of torch SDPA on [b, a, s, d] tensors; this is the same call."""
import torch
import torch.nn.functional as F


def fullattn_torch_cpu(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, attn_mask=None) -> torch.Tensor:
    """q [b,a,s,d], k/v [b,a,s1,d] CPU tensors -> [b,a,s,d] (attn.py:101-106, :153)."""
    assert not q.is_cuda
    if attn_mask is not None and attn_mask.dtype != torch.bool:
        attn_mask = attn_mask.to(q.dtype)
    return F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, dropout_p=0.0, is_causal=False)
