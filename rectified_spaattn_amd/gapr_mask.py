"""GAPR mask: `estimate_pr_gain` with the reference's signature (gapr_mask.py:4-42), on the HIP library."""
import ctypes

import torch

from . import _core, _lib


def estimate_pr_gain(Q_blocks, K_blocks, q_pools, k_pools, attention_scores):
    """Q_blocks/K_blocks [B,H,N,128,d] (bf16/fp16 device tensors), q_pools/k_pools [B,H,N,d], attention_scores
    [B,H,NQ,NK] (unscaled pooled scores).  Returns bool [B,H,NQ,NK]: True where the pooled-score gain does NOT
    exceed the pooling error, i.e. ~gapr_mask, exactly what the reference returns.

    err = |mean_rows|Q - q_pool| . k_pool| + |q_pool . mean_rows|K - k_pool||  vs  |score|   (the common factor
    IQ*JK of gapr_mask.py:27,33,38 cancels); statistics in fp32 under the numeric contract of oracle/rsa_oracle.c."""
    _core._require_device(Q_blocks, K_blocks, q_pools, k_pools, attention_scores)
    B, H, NQ, IQ, d = Q_blocks.shape
    NK, JK = K_blocks.shape[2], K_blocks.shape[3]
    if IQ != _lib.BLOCK or JK != _lib.BLOCK:
        raise NotImplementedError("estimate_pr_gain on the HIP path needs 128-token blocks")
    BH = B * H
    qb, kb = Q_blocks.contiguous(), K_blocks.contiguous()
    qp = q_pools.reshape(BH, NQ, d).float().contiguous()
    kp = k_pools.reshape(BH, NK, d).float().contiguous()
    sc = attention_scores.reshape(BH, NQ, NK).float().contiguous()
    s_q = torch.empty(2 * BH * NQ * d, dtype=torch.float32, device=qb.device)
    s_k = torch.empty(2 * BH * NK * d, dtype=torch.float32, device=qb.device)
    out = torch.empty(BH, NQ, NK, dtype=torch.uint8, device=qb.device)
    vp = ctypes.c_void_p
    with torch.cuda.device(qb.device):
        _lib.check(_lib.lib().rsa_estimate_pr_gain(BH, NQ, NK, d, _core.dtype_code(qb.dtype), vp(qb.data_ptr()),
                                                   vp(kb.data_ptr()), vp(qp.data_ptr()), vp(kp.data_ptr()),
                                                   vp(sc.data_ptr()), vp(s_q.data_ptr()), vp(s_k.data_ptr()),
                                                   vp(out.data_ptr()), _core._stream()), "rsa_estimate_pr_gain")
    return out.view(B, H, NQ, NK).bool()
