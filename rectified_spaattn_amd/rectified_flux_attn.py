"""Flux.1-dev variant (reference: rectified_flux_attn.py).  Sequence layout [image (Hilbert order) | text]."""
from typing import Optional

import torch
import torch.nn.functional as F

from . import _operator as op
from .attn import fullattn
from .gapr_mask import estimate_pr_gain  # noqa: F401


def block_sparse_attention_combined(query, key, value, attn_mask, top_k, block_size_M=128, block_size_N=128,
                                    cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None,
                                    prob_threshold=0.5, block_neighbor_list=None, text_length=256,
                                    shape_xfuse=False, qkv_fp8=None):
    """[B,H,S,D] x3 -> [B,S,H*D]; the last `text_length` tokens are text: kept by every visual row, scored
    token-wise for IPAR, and their own rows get exact attention (reference :282-376)."""
    return op.run("flux", query, key, value, top_k, prob_threshold, block_neighbor_list, shape_xfuse,
                  cu_seqlens_q=cu_seqlens_q, cu_seqlens_kv=cu_seqlens_kv, text_length=text_length,
                  block_size_M=block_size_M, block_size_N=block_size_N, qkv_fp8=qkv_fp8)


def rectified_block_sparse_attention(query, key, value, attn_mask, top_k, block_size_M=128, block_size_N=128,
                                     cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None,
                                     block_neighbor_list=None, shape_xfuse=False, p_remain_rates=0.5,
                                     text_length=256, qkv_fp8=None):
    return block_sparse_attention_combined(query, key, value, attn_mask, top_k, block_size_M, block_size_N,
                                           cu_seqlens_q, cu_seqlens_kv, max_seqlen_q, max_seqlen_kv,
                                           prob_threshold=p_remain_rates, block_neighbor_list=block_neighbor_list,
                                           text_length=text_length, shape_xfuse=shape_xfuse, qkv_fp8=qkv_fp8)


class RectifiedFluxSpaAttnProcessor2_0:
    """Reference :408-542.  Sparse on every layer except processor ids 37..56 (dense warm-up band, :493).
    Deviation on purpose: the dense branch honours self.mode ("torch"/"vanilla" run on
    CPU tensors) instead of being hard-wired to "flash"; on device all dense modes are the same HIP kernel."""

    def __init__(self, mode, select_block_num, block_neighbor_list, p_remain_rates, processor_id=0, text_length=256):
        if not hasattr(F, "scaled_dot_product_attention"):
            raise ImportError("FluxAttnProcessor2_0 requires PyTorch 2.0. To use it, please upgrade PyTorch to 2.0.")
        self.mode = mode
        self.select_block_num = select_block_num
        self.block_neighbor_list = block_neighbor_list
        self.p_remain_rates = p_remain_rates
        self.current_step = 0
        self.processor_id = processor_id
        # K5 / dense-kernel operand precision of THIS processor (None = process default, see set_qkv_fp8 / set_dense_fp8)
        self.qkv_fp8 = None
        self.dense_fp8 = None
        self.text_length = text_length

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, image_rotary_emb=None):
        dual = encoder_hidden_states is not None
        norms = [attn.norm_q, attn.norm_k] + ([attn.norm_added_q, attn.norm_added_k] if dual else [])
        if op.fused_qk_ok(hidden_states, attn.heads, norms, image_rotary_emb):
            # RMSNorm + RoPE (over the whole [image | text] sequence) fused, written into the concat buffers
            from . import glue
            Bq, S_v, _ = hidden_states.shape
            n_enc = encoder_hidden_states.shape[1] if dual else 0
            hd = hidden_states.shape[-1] // attn.heads
            qbuf = torch.empty((Bq, S_v + n_enc, attn.heads, hd), dtype=hidden_states.dtype, device=hidden_states.device)
            kbuf = torch.empty_like(qbuf)
            rot_v = rot_t = None
            if image_rotary_emb is not None:  # rows [0, S_v) for the image part, [S_v, S_v + n_enc) for the text part
                rot_v = (image_rotary_emb[0][:S_v], image_rotary_emb[1][:S_v])
                rot_t = (image_rotary_emb[0][S_v:], image_rotary_emb[1][S_v:])
            glue.qk_norm_rope(attn.to_q(hidden_states), attn.heads, op.norm_args(attn.norm_q), rot_v, S_v,
                              out=qbuf[:, :S_v])
            glue.qk_norm_rope(attn.to_k(hidden_states), attn.heads, op.norm_args(attn.norm_k), rot_v, S_v,
                              out=kbuf[:, :S_v])
            v_all = attn.to_v(hidden_states)
            if dual:
                glue.qk_norm_rope(attn.add_q_proj(encoder_hidden_states), attn.heads,
                                  op.norm_args(attn.norm_added_q), rot_t, n_enc, out=qbuf[:, S_v:])
                glue.qk_norm_rope(attn.add_k_proj(encoder_hidden_states), attn.heads,
                                  op.norm_args(attn.norm_added_k), rot_t, n_enc, out=kbuf[:, S_v:])
                v_all = torch.cat([v_all, attn.add_v_proj(encoder_hidden_states)], dim=1)
            q, k, v = qbuf.transpose(1, 2), kbuf.transpose(1, 2), op.split_heads(v_all, attn.heads)
        else:
            q = op.split_heads(attn.to_q(hidden_states), attn.heads)
            k = op.split_heads(attn.to_k(hidden_states), attn.heads)
            v = op.split_heads(attn.to_v(hidden_states), attn.heads)
            if attn.norm_q is not None:
                q = attn.norm_q(q)
            if attn.norm_k is not None:
                k = attn.norm_k(k)
            if dual:  # double-stream block: text goes LAST ("Jenga" order, :476-478)
                eq = op.split_heads(attn.add_q_proj(encoder_hidden_states), attn.heads)
                ek = op.split_heads(attn.add_k_proj(encoder_hidden_states), attn.heads)
                ev = op.split_heads(attn.add_v_proj(encoder_hidden_states), attn.heads)
                if attn.norm_added_q is not None:
                    eq = attn.norm_added_q(eq)
                if attn.norm_added_k is not None:
                    ek = attn.norm_added_k(ek)
                q, k, v = torch.cat([q, eq], 2), torch.cat([k, ek], 2), torch.cat([v, ev], 2)
            if image_rotary_emb is not None:
                q, k = op.rotary(q, image_rotary_emb), op.rotary(k, image_rotary_emb)

        B, H, S_q, D = q.shape
        S_k = k.shape[2]
        s_k = op.valid_keys(attention_mask, S_k)
        cu_q, cu_kv = [0, S_q, S_q], [0, s_k, S_k]
        sparse_layer = self.processor_id < 37 or self.processor_id >= 57
        if self.mode == "sparse" and sparse_layer:
            out = rectified_block_sparse_attention(q, k, v, attn_mask=attention_mask, top_k=self.select_block_num,
                                                   cu_seqlens_q=cu_q, cu_seqlens_kv=cu_kv, max_seqlen_q=S_q,
                                                   max_seqlen_kv=S_k, block_neighbor_list=self.block_neighbor_list,
                                                   p_remain_rates=self.p_remain_rates, text_length=self.text_length, qkv_fp8=self.qkv_fp8)
        else:
            dense_mode = self.mode if self.mode in ("torch", "vanilla") else "flash"
            out = fullattn(q, k, v, mode=dense_mode, drop_rate=0.0, attn_mask=attention_mask, causal=False,
                           cu_seqlens_q=cu_q, cu_seqlens_kv=cu_kv, max_seqlen_q=S_q, max_seqlen_kv=S_k, batch_size=B, dense_fp8=self.dense_fp8)
            out = out.transpose(1, 2).reshape(B, S_q, -1)
        out = out.to(q.dtype)
        self.current_step = (self.current_step + 1) % 50
        if encoder_hidden_states is None:
            return out
        n_txt = encoder_hidden_states.shape[1]
        out, enc = out[:, :-n_txt], out[:, -n_txt:]
        out = attn.to_out[1](attn.to_out[0](out))
        return out, attn.to_add_out(enc)
