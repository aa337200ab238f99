"""Wan2.2 processors (reference: rectified_wan22_attn.py) -- same operator as Wan2.1, new-style diffusers
attention module (fused projections, cos/sin RoPE, heads kept in dim 2 until the attention call)."""
import torch

from . import _operator as op
from .rectified_wan21_attn import _WanProcessorBase, rectified_block_sparse_attention  # noqa: F401


def _qkv_projections(attn, hidden_states, encoder_hidden_states):
    """diffusers' transformer_wan._get_qkv_projections when available; otherwise the same logic: fused to_qkv /
    to_kv when the module was fused, separate to_q/to_k/to_v else."""
    try:
        from diffusers.models.transformers.transformer_wan import _get_qkv_projections
        return _get_qkv_projections(attn, hidden_states, encoder_hidden_states)
    except ImportError:
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        if getattr(attn, "fused_projections", False):
            if getattr(attn, "is_cross_attention", False):
                q = attn.to_q(hidden_states)
                k, v = attn.to_kv(encoder_hidden_states).chunk(2, dim=-1)
            else:
                q, k, v = attn.to_qkv(hidden_states).chunk(3, dim=-1)
            return q, k, v
        return attn.to_q(hidden_states), attn.to_k(encoder_hidden_states), attn.to_v(encoder_hidden_states)


def _added_kv_projections(attn, enc_img):
    try:
        from diffusers.models.transformers.transformer_wan import _get_added_kv_projections
        return _get_added_kv_projections(attn, enc_img)
    except ImportError:
        if getattr(attn, "fused_projections", False):
            return attn.to_added_kv(enc_img).chunk(2, dim=-1)
        return attn.add_k_proj(enc_img), attn.add_v_proj(enc_img)


def _cos_sin_rope(x, freqs_cos, freqs_sin):
    """x [B,S,H,D]: even/odd channel pairs rotated by (cos, sin) tables (reference :54-66)."""
    x1, x2 = x.unflatten(-1, (-1, 2)).unbind(-1)
    cos, sin = freqs_cos[..., 0::2], freqs_sin[..., 1::2]
    out = torch.empty_like(x)
    out[..., 0::2] = x1 * cos - x2 * sin
    out[..., 1::2] = x1 * sin + x2 * cos
    return out.type_as(x)


class _Wan22Base(_WanProcessorBase):
    def __init__(self, mode, select_block_num, block_neighbor_list, p_remain_rates, processor_id=0,
                 first_frame_blocks=0, warm_steps=0):
        super().__init__(mode, select_block_num, block_neighbor_list, p_remain_rates, processor_id,
                         first_frame_blocks)
        self.warm_steps = warm_steps
        self._attention_backend = None

    def _qkv(self, attn, hidden_states, encoder_hidden_states, rotary_emb):
        q, k, v = _qkv_projections(attn, hidden_states, encoder_hidden_states)
        if op.fused_heads_ok(q, attn.heads, (attn.norm_q,), rotary_emb) and \
                op.fused_heads_ok(k, attn.heads, (attn.norm_k,), rotary_emb):
            from . import glue   # RMSNorm across heads + cos/sin rotation + head split in one pass per tensor
            q = glue.norm_rope_across_heads(q, attn.heads, glue.norm_params(attn.norm_q), rotary_emb)
            k = glue.norm_rope_across_heads(k, attn.heads, glue.norm_params(attn.norm_k), rotary_emb)
            return q, k, op.split_heads(v, attn.heads)
        q, k = attn.norm_q(q), attn.norm_k(k)
        q, k, v = (x.unflatten(2, (attn.heads, -1)) for x in (q, k, v))
        if rotary_emb is not None:
            q, k = _cos_sin_rope(q, *rotary_emb), _cos_sin_rope(k, *rotary_emb)
        return q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)

    def _image_kv(self, attn, enc_img):
        k_img, v_img = _added_kv_projections(attn, enc_img)
        k_img = attn.norm_added_k(k_img)
        return op.split_heads(k_img, attn.heads), op.split_heads(v_img, attn.heads)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, rotary_emb=None):
        # the new-style module passes encoder_hidden_states=None for self-attention and expects the projection
        # helper to fall back to hidden_states; _WanProcessorBase does the same substitution.
        return super().__call__(attn, hidden_states, encoder_hidden_states, attention_mask, rotary_emb)


class RectifiedWanTI2VSpaAttnProcessor2_0(_Wan22Base):
    """Reference :15-163 (TI2V-5B): no warm_steps argument; sparse from layer 2 and step counter 10, wrap 100."""

    def __init__(self, mode, select_block_num, block_neighbor_list, p_remain_rates, processor_id=0,
                 first_frame_blocks=0):
        super().__init__(mode, select_block_num, block_neighbor_list, p_remain_rates, processor_id,
                         first_frame_blocks, warm_steps=0)

    def _use_sparse(self):
        return self.processor_id >= 2 and self.current_step >= 10


class RectifiedWanT2VSpaAttnProcessor2_0(_Wan22Base):
    """Reference :166-288 (T2V-A14B, two 40-layer experts): dense on layers 0, 1, 40, 41 and for the first
    warm_steps calls; step counter wraps at 80."""
    _wrap = 80

    def _use_sparse(self):
        return self.processor_id not in (0, 1, 40, 41) and self.current_step >= self.warm_steps


class RectifiedWanI2VSpaAttnProcessor2_0(RectifiedWanT2VSpaAttnProcessor2_0):
    """Reference :291-413 (I2V-A14B): same gating as the Wan2.2 T2V processor."""
