"""HunyuanVideo variant: operator + diffusers attention processor (reference: rectified_hunyuan_attn.py).

Sequence layout [visual (Hilbert order) | text padded to 256]; attention_mask marks the valid prefix."""
from typing import Optional

import torch
import torch.nn.functional as F

from . import _operator as op
from .attn import fullattn
from .gapr_mask import estimate_pr_gain  # noqa: F401  (re-exported like the reference module)


def block_sparse_attention_combined(query, key, value, attn_mask, top_k, block_size_M=128, block_size_N=128,
                                    cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None,
                                    prob_threshold=0.5, block_neighbor_list=None, shape_xfuse=False, qkv_fp8=None):
    """[B,H,S,D] x3 -> [B,S,H*D].  Visual query blocks: rectified block-sparse attention; text rows: exact
    attention over the valid keys; padded text rows: 0 (reference :283-389).  cu_seqlens_q = [0, num_true, S].
    Unlike the reference (:307-308) key/value are NOT modified in place; masked rows are predicated to zero."""
    return op.run("hunyuan", query, key, value, top_k, prob_threshold, block_neighbor_list, shape_xfuse,
                  cu_seqlens_q=cu_seqlens_q, cu_seqlens_kv=cu_seqlens_kv, block_size_M=block_size_M,
                  block_size_N=block_size_N, qkv_fp8=qkv_fp8)


def rectified_block_sparse_attention(query, key, value, attn_mask, top_k, block_size_M=128, block_size_N=128,
                                     cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None,
                                     block_neighbor_list=None, shape_xfuse=False, p_remain_rates=0.5, qkv_fp8=None):
    """Public alias with the reference's keyword names (:393-417)."""
    return block_sparse_attention_combined(query, key, value, attn_mask, top_k, block_size_M, block_size_N,
                                           cu_seqlens_q, cu_seqlens_kv, max_seqlen_q, max_seqlen_kv,
                                           prob_threshold=p_remain_rates, block_neighbor_list=block_neighbor_list,
                                           shape_xfuse=shape_xfuse, qkv_fp8=qkv_fp8)


class RectifiedHunyuanVideoSpaAttnProcessor2_0:
    """Drop-in for the reference processor (:419-545): positional ctor (mode, select_block_num,
    block_neighbor_list, p_remain_rates, processor_id), same __call__ keywords, step counter wrapping at 50."""

    def __init__(self, mode, select_block_num, block_neighbor_list, p_remain_rates, processor_id=0):
        if not hasattr(F, "scaled_dot_product_attention"):
            raise ImportError("HunyuanVideoAttnProcessor2_0 requires PyTorch 2.0. To use it, please upgrade "
                              "PyTorch to 2.0.")
        self.mode = mode
        self.select_block_num = select_block_num
        self.block_neighbor_list = block_neighbor_list
        self.p_remain_rates = p_remain_rates
        self.current_step = 0
        self.processor_id = processor_id
        # K5 / dense-kernel operand precision of THIS processor (None = process default, see set_qkv_fp8 / set_dense_fp8)
        self.qkv_fp8 = None
        self.dense_fp8 = None

    def __call__(self, attn, hidden_states: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None, image_rotary_emb=None,
                 num_true: Optional[int] = None):
        """num_true (optional, not in the reference): the count of valid keys = attention_mask.sum(), for callers that
        computed it once per forward; without it the count is taken from the mask (one host sync per mask object)."""
        single_stream = attn.add_q_proj is None and encoder_hidden_states is not None
        n_txt = encoder_hidden_states.shape[1] if encoder_hidden_states is not None else 0
        if single_stream:  # single-stream blocks project the concatenated sequence
            hidden_states = torch.cat([hidden_states, encoder_hidden_states], dim=1)
        dual = attn.add_q_proj is not None and encoder_hidden_states is not None
        norms = [attn.norm_q, attn.norm_k] + ([attn.norm_added_q, attn.norm_added_k] if dual else [])
        if op.fused_qk_ok(hidden_states, attn.heads, norms, image_rotary_emb):
            # one pass per tensor: RMSNorm + RoPE, written straight into the [visual | text] buffers
            from . import glue
            Bq, S_v, _ = hidden_states.shape
            S_all = S_v + (n_txt if dual else 0)
            hd = hidden_states.shape[-1] // attn.heads
            qbuf = torch.empty((Bq, S_all, attn.heads, hd), dtype=hidden_states.dtype, device=hidden_states.device)
            kbuf = torch.empty_like(qbuf)
            rope_tokens = S_v - n_txt if single_stream else S_v
            glue.qk_norm_rope(attn.to_q(hidden_states), attn.heads, op.norm_args(attn.norm_q), image_rotary_emb,
                              rope_tokens, out=qbuf[:, :S_v])
            glue.qk_norm_rope(attn.to_k(hidden_states), attn.heads, op.norm_args(attn.norm_k), image_rotary_emb,
                              rope_tokens, out=kbuf[:, :S_v])
            v_all = attn.to_v(hidden_states)
            if dual:
                glue.qk_norm_rope(attn.add_q_proj(encoder_hidden_states), attn.heads,
                                  op.norm_args(attn.norm_added_q), None, 0, out=qbuf[:, S_v:])
                glue.qk_norm_rope(attn.add_k_proj(encoder_hidden_states), attn.heads,
                                  op.norm_args(attn.norm_added_k), None, 0, out=kbuf[:, S_v:])
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
            if image_rotary_emb is not None:  # RoPE on the visual part only
                if single_stream:
                    q = torch.cat([op.rotary(q[:, :, :-n_txt], image_rotary_emb), q[:, :, -n_txt:]], dim=2)
                    k = torch.cat([op.rotary(k[:, :, :-n_txt], image_rotary_emb), k[:, :, -n_txt:]], dim=2)
                else:
                    q, k = op.rotary(q, image_rotary_emb), op.rotary(k, image_rotary_emb)
            if dual:  # dual-stream: text appended last
                eq = op.split_heads(attn.add_q_proj(encoder_hidden_states), attn.heads)
                ek = op.split_heads(attn.add_k_proj(encoder_hidden_states), attn.heads)
                ev = op.split_heads(attn.add_v_proj(encoder_hidden_states), attn.heads)
                if attn.norm_added_q is not None:
                    eq = attn.norm_added_q(eq)
                if attn.norm_added_k is not None:
                    ek = attn.norm_added_k(ek)
                q, k, v = torch.cat([q, eq], 2), torch.cat([k, ek], 2), torch.cat([v, ev], 2)

        B, H, S, D = q.shape
        num_true = op.valid_keys(attention_mask, S, num_true)
        cu = [0, num_true, S]
        if self.mode == "sparse":
            out = rectified_block_sparse_attention(q, k, v, attn_mask=attention_mask, top_k=self.select_block_num,
                                                   cu_seqlens_q=cu, cu_seqlens_kv=cu, max_seqlen_q=S,
                                                   max_seqlen_kv=S, block_neighbor_list=self.block_neighbor_list,
                                                   p_remain_rates=self.p_remain_rates, qkv_fp8=self.qkv_fp8)
        elif self.mode in ("flash", "torch", "vanilla"):
            out = fullattn(q, k, v, mode=self.mode, drop_rate=0.0, attn_mask=attention_mask, causal=False,
                           cu_seqlens_q=cu, cu_seqlens_kv=cu, max_seqlen_q=S, max_seqlen_kv=S, batch_size=B, dense_fp8=self.dense_fp8)
            out = out.transpose(1, 2).reshape(B, S, -1)
        else:
            raise ImportError("Undefined Attention Processor! Just support sparse, flash, torch, vanilla.")

        if encoder_hidden_states is not None:
            out, encoder_hidden_states = out[:, :-n_txt], out[:, -n_txt:]
            if getattr(attn, "to_out", None) is not None:
                out = attn.to_out[1](attn.to_out[0](out))
            if getattr(attn, "to_add_out", None) is not None:
                encoder_hidden_states = attn.to_add_out(encoder_hidden_states)
        self.current_step = (self.current_step + 1) % 50
        return out, encoder_hidden_states
