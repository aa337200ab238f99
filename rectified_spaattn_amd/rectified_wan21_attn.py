"""Wan2.1 variant (reference: rectified_wan21_attn.py): visual-only sequence, zero-padded to a multiple of
128, optional first-frame square; T2V and I2V processors (self-attention sparse, cross-attention dense)."""
from typing import Optional

import torch
import torch.nn.functional as F

from . import _operator as op
from .attn import fullattn
from .gapr_mask import estimate_pr_gain  # noqa: F401


def block_sparse_attention_combined(query, key, value, attn_mask, top_k, block_size_M=128, block_size_N=128,
                                    cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None,
                                    prob_threshold=0.5, block_neighbor_list=None, shape_xfuse=False,
                                    first_frame_blocks=None, qkv_fp8=None):
    """[B,H,S,D] x3 -> [B,S,H*D] (reference :276-357).  S need not be a multiple of 128: the tail block is
    treated as zero-padded (its pooled statistics include the zeros, as in the reference :299-302)."""
    return op.run("wan", query, key, value, top_k, prob_threshold, block_neighbor_list, shape_xfuse,
                  first_frame_blocks=first_frame_blocks, block_size_M=block_size_M, block_size_N=block_size_N, qkv_fp8=qkv_fp8)


def rectified_block_sparse_attention(query, key, value, attn_mask, top_k, block_size_M=128, block_size_N=128,
                                     cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None,
                                     block_neighbor_list=None, shape_xfuse=False, p_remain_rates=0.5,
                                     first_frame_blocks=None, qkv_fp8=None):
    return block_sparse_attention_combined(query, key, value, attn_mask, top_k, block_size_M, block_size_N,
                                           cu_seqlens_q, cu_seqlens_kv, max_seqlen_q, max_seqlen_kv,
                                           prob_threshold=p_remain_rates, block_neighbor_list=block_neighbor_list,
                                           shape_xfuse=shape_xfuse, first_frame_blocks=first_frame_blocks, qkv_fp8=qkv_fp8)


def _complex_rope(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """Wan2.1 RoPE: pairs of channels rotated by complex frequencies, in fp64 (fp32 on MPS) (:434-438)."""
    wide = torch.float32 if x.device.type == "mps" else torch.float64
    xc = torch.view_as_complex(x.to(wide).unflatten(3, (-1, 2)))
    return torch.view_as_real(xc * freqs).flatten(3, 4).type_as(x)


class _WanProcessorBase:
    """Shared body of the Wan processors.  Sub-classes define `_use_sparse()` (the warm-up gate) and `_wrap`
    (step-counter period)."""
    _wrap = 100
    _name = "WanAttnProcessor2_0"

    def __init__(self, mode, select_block_num, block_neighbor_list, p_remain_rates, processor_id=0,
                 first_frame_blocks=0):
        if not hasattr(F, "scaled_dot_product_attention"):
            raise ImportError(f"{self._name} requires PyTorch 2.0. To use it, please upgrade PyTorch to 2.0.")
        self.mode = mode
        self.select_block_num = select_block_num
        self.block_neighbor_list = block_neighbor_list
        self.p_remain_rates = p_remain_rates
        self.current_step = 0
        self.processor_id = processor_id
        # K5 / dense-kernel operand precision of THIS processor (None = process default, see set_qkv_fp8 / set_dense_fp8)
        self.qkv_fp8 = None
        self.dense_fp8 = None
        self.first_frame_blocks = first_frame_blocks

    def _use_sparse(self) -> bool:
        raise NotImplementedError

    # -- projections / rope differ between the Wan2.1 and Wan2.2 diffusers modules --
    def _qkv(self, attn, hidden_states, encoder_hidden_states, rotary_emb):
        q = attn.to_q(hidden_states)
        k = attn.to_k(encoder_hidden_states)
        v = attn.to_v(encoder_hidden_states)
        if op.fused_heads_ok(q, attn.heads, (attn.norm_q,), rotary_emb) and \
                op.fused_heads_ok(k, attn.heads, (attn.norm_k,), rotary_emb):
            # one pass per tensor: RMSNorm across heads + the fp64 rotation + the head split (glue.norm_rope_across_heads)
            from . import glue
            q = glue.norm_rope_across_heads(q, attn.heads, glue.norm_params(attn.norm_q), rotary_emb)
            k = glue.norm_rope_across_heads(k, attn.heads, glue.norm_params(attn.norm_k), rotary_emb)
            return q, k, op.split_heads(v, attn.heads)
        if attn.norm_q is not None:
            q = attn.norm_q(q)
        if attn.norm_k is not None:
            k = attn.norm_k(k)
        q, k, v = (op.split_heads(x, attn.heads) for x in (q, k, v))
        if rotary_emb is not None:
            q, k = _complex_rope(q, rotary_emb), _complex_rope(k, rotary_emb)
        return q, k, v

    def _image_kv(self, attn, enc_img):
        k_img = attn.norm_added_k(attn.add_k_proj(enc_img))
        v_img = attn.add_v_proj(enc_img)
        return op.split_heads(k_img, attn.heads), op.split_heads(v_img, attn.heads)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, rotary_emb=None):
        enc_img = None
        if attn.add_k_proj is not None:  # I2V cross-attention: [image context | 512 text tokens]
            n_img = encoder_hidden_states.shape[1] - 512
            enc_img, encoder_hidden_states = encoder_hidden_states[:, :n_img], encoder_hidden_states[:, n_img:]
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        q, k, v = self._qkv(attn, hidden_states, encoder_hidden_states, rotary_emb)

        out_img = None
        if enc_img is not None:  # 257 CLIP tokens: tiny, stays on torch SDPA (out of the hot path)
            k_img, v_img = self._image_kv(attn, enc_img)
            out_img = F.scaled_dot_product_attention(q, k_img, v_img, attn_mask=None, dropout_p=0.0, is_causal=False)
            out_img = out_img.transpose(1, 2).flatten(2, 3).type_as(q)

        B, H, S_q, D = q.shape
        S_k = k.shape[2]
        s_k = op.valid_keys(attention_mask, S_k)
        cu_q, cu_kv = [0, S_q, S_q], [0, s_k, S_k]
        if self.mode == "sparse" and self._use_sparse():
            out = rectified_block_sparse_attention(q, k, v, attn_mask=attention_mask, top_k=self.select_block_num,
                                                   cu_seqlens_q=cu_q, cu_seqlens_kv=cu_kv, max_seqlen_q=S_q,
                                                   max_seqlen_kv=S_k, block_neighbor_list=self.block_neighbor_list,
                                                   p_remain_rates=self.p_remain_rates,
                                                   first_frame_blocks=self.first_frame_blocks, qkv_fp8=self.qkv_fp8)
        elif self.mode in ("sparse", "flash", "torch", "vanilla"):
            dense_mode = "flash" if self.mode == "sparse" else self.mode  # warm-up layers/steps run dense
            out = fullattn(q, k, v, mode=dense_mode, drop_rate=0.0, attn_mask=attention_mask, causal=False,
                           cu_seqlens_q=cu_q, cu_seqlens_kv=cu_kv, max_seqlen_q=S_q, max_seqlen_kv=S_k, batch_size=B, dense_fp8=self.dense_fp8)
            out = out.transpose(1, 2).reshape(B, S_q, -1)
        else:
            raise ImportError("Undefined Attention Processor! Just support sparse, flash, torch, vanilla.")
        out = out.type_as(q)
        self.current_step = (self.current_step + 1) % self._wrap
        if out_img is not None:
            out = out + out_img
        return attn.to_out[1](attn.to_out[0](out))


class RectifiedWanT2VSpaAttnProcessor2_0(_WanProcessorBase):
    """Reference :389-509: sparse from layer 2 on and from step counter 10 on (2 calls per step: cond/uncond)."""

    def _use_sparse(self):
        return self.processor_id >= 2 and self.current_step >= 10


class RectifiedWanI2VSpaAttnProcessor2_0(_WanProcessorBase):
    """Reference :512-632: sparse from layer 2 on, no step warm-up (:591)."""

    def _use_sparse(self):
        return self.processor_id >= 2
