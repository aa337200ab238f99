"""CogVideoX1.5 variant (reference: rectified_cogvideo_attn.py): [visual | text] with zero padding to x128,
every text block kept, head_dim 64."""
import torch
import torch.nn.functional as F

from . import _operator as op
from .attn import fullattn
from .gapr_mask import estimate_pr_gain  # noqa: F401


def block_sparse_attention_combined(query, key, value, attn_mask, top_k, block_size_M=128, block_size_N=128,
                                    cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None,
                                    prob_threshold=0.5, block_neighbor_list=None, text_length=256,
                                    shape_xfuse=False, qkv_fp8=None):
    """[B,H,S,D] x3 -> [B,S,H*D] (reference :282-378)."""
    return op.run("cogvideo", query, key, value, top_k, prob_threshold, block_neighbor_list, shape_xfuse,
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


class RectifiedCogVideoXVideoSpaAttnProcessor2_0:
    """Reference :410-523: sparse once the step counter reaches 5; dense otherwise (any mode)."""

    def __init__(self, mode, select_block_num, block_neighbor_list, p_remain_rates, processor_id=0):
        if not hasattr(F, "scaled_dot_product_attention"):
            raise ImportError("CogVideoXAttnProcessor requires PyTorch 2.0, to use it, please upgrade PyTorch to 2.0.")
        self.mode = mode
        self.select_block_num = select_block_num
        self.block_neighbor_list = block_neighbor_list
        self.p_remain_rates = p_remain_rates
        self.current_step = 0
        self.processor_id = processor_id
        # K5 / dense-kernel operand precision of THIS processor (None = process default, see set_qkv_fp8 / set_dense_fp8)
        self.qkv_fp8 = None
        self.dense_fp8 = None

    def __call__(self, attn, hidden_states, encoder_hidden_states, attention_mask=None, image_rotary_emb=None):
        n_txt = encoder_hidden_states.size(1)
        x = torch.cat([hidden_states, encoder_hidden_states], dim=1)  # visual first, text last
        B, S, _ = x.shape
        if attention_mask is not None:
            attention_mask = attn.prepare_attention_mask(attention_mask, S, B)
            attention_mask = attention_mask.view(B, attn.heads, -1, attention_mask.shape[-1])
        q, k, v = attn.to_q(x), attn.to_k(x), attn.to_v(x)
        if op.fused_qk_ok(q, attn.heads, (attn.norm_q, attn.norm_k), image_rotary_emb):
            # per-head LayerNorm + RoPE on the visual tokens + head split in one pass per tensor (rsa_qk_layernorm_rope)
            from . import glue
            rope_k = image_rotary_emb if not attn.is_cross_attention else None
            q = glue.qk_norm_rope(q, attn.heads, op.norm_args(attn.norm_q), image_rotary_emb, S - n_txt)
            k = glue.qk_norm_rope(k, attn.heads, op.norm_args(attn.norm_k), rope_k, S - n_txt)
            v = op.split_heads(v, attn.heads)
        else:
            q, k, v = (op.split_heads(t, attn.heads) for t in (q, k, v))
            if attn.norm_q is not None:
                q = attn.norm_q(q)
            if attn.norm_k is not None:
                k = attn.norm_k(k)
            if image_rotary_emb is not None:  # RoPE on the visual tokens only
                q = torch.cat([op.rotary(q[:, :, :-n_txt], image_rotary_emb), q[:, :, -n_txt:]], dim=2)
                if not attn.is_cross_attention:
                    k = torch.cat([op.rotary(k[:, :, :-n_txt], image_rotary_emb), k[:, :, -n_txt:]], dim=2)
        S_k = k.shape[2]
        s_k = op.valid_keys(attention_mask, S_k)
        cu_q, cu_kv = [0, S, S * B], [0, s_k, S_k * B]
        if self.mode == "sparse" and self.current_step >= 5:
            out = rectified_block_sparse_attention(q, k, v, attn_mask=attention_mask, top_k=self.select_block_num,
                                                   cu_seqlens_q=cu_q, cu_seqlens_kv=cu_kv, max_seqlen_q=S,
                                                   max_seqlen_kv=S_k, block_neighbor_list=self.block_neighbor_list,
                                                   p_remain_rates=self.p_remain_rates, text_length=n_txt, qkv_fp8=self.qkv_fp8)
        else:
            dense_mode = self.mode if self.mode in ("torch", "vanilla") else "flash"
            out = fullattn(q, k, v, mode=dense_mode, drop_rate=0.0, attn_mask=attention_mask, causal=False,
                           cu_seqlens_q=cu_q, cu_seqlens_kv=cu_kv, max_seqlen_q=S, max_seqlen_kv=S_k, batch_size=B, dense_fp8=self.dense_fp8)
            out = out.transpose(1, 2).reshape(B, S, -1)
        out = out.to(q.dtype)
        self.current_step = (self.current_step + 1) % 50
        out = attn.to_out[1](attn.to_out[0](out))
        return out[:, : S - n_txt], out[:, S - n_txt:]
