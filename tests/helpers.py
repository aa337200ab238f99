"""Fake diffusers `Attention` modules and deterministic inputs for the processor tests (shared by
tests/golden/make_golden.py and the tests, so fixtures and tests build identical modules)."""
import types

import numpy as np
import torch
import torch.nn as nn

from rectified_spaattn_amd import synth


def _linear(seed, stream, din, dout, scale=0.06):
    lin = nn.Linear(din, dout, bias=True)
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy((synth.normal(seed, stream, (dout, din)) * scale).astype(np.float32)))
        lin.bias.copy_(torch.from_numpy((synth.normal(seed, stream + 50, (dout,)) * 0.02).astype(np.float32)))
    return lin


class RMS(nn.Module):
    """diffusers.models.normalization.RMSNorm, operation by operation (weight / eps attributes included)."""

    def __init__(self, d, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.linspace(0.8, 1.2, d))
        self.eps = eps
        self.bias = None

    def forward(self, x):
        input_dtype = x.dtype
        variance = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        x = x * torch.rsqrt(variance + self.eps)
        if self.weight.dtype in (torch.float16, torch.bfloat16):
            x = x.to(self.weight.dtype)
        return (x * self.weight).to(input_dtype)


def fake_attn(seed, heads, head_dim, added=True, norms=True, wan=False):
    dim = heads * head_dim
    a = types.SimpleNamespace()
    a.heads = heads
    a.to_q, a.to_k, a.to_v = (_linear(seed, i, dim, dim) for i in (1, 2, 3))
    a.to_out = nn.ModuleList([_linear(seed, 4, dim, dim), nn.Identity()])
    if wan:  # Wan norms act on the full hidden dim before the head split
        a.norm_q, a.norm_k = (RMS(dim), RMS(dim)) if norms else (None, None)
        a.add_k_proj = None
        a.add_q_proj = None
    else:
        a.norm_q, a.norm_k = (RMS(head_dim), RMS(head_dim)) if norms else (None, None)
        if added:
            a.add_q_proj, a.add_k_proj, a.add_v_proj = (_linear(seed, i, dim, dim) for i in (5, 6, 7))
            a.norm_added_q, a.norm_added_k = (RMS(head_dim), RMS(head_dim)) if norms else (None, None)
            a.to_add_out = _linear(seed, 8, dim, dim)
        else:
            a.add_q_proj = a.add_k_proj = a.add_v_proj = None
            a.norm_added_q = a.norm_added_k = None
            a.to_add_out = None
    a.is_cross_attention = False
    a.modules_ = [m for m in vars(a).values() if isinstance(m, nn.Module)]
    return a


def attn_to(a, device=None, dtype=None):
    for m in a.modules_:
        m.to(device=device, dtype=dtype)
    return a


def hidden(seed, stream, B, S, dim):
    return torch.from_numpy(synth.normal(seed, stream, (B, S, dim)).astype(np.float32))


def rope_tables(S, head_dim):
    """(cos, sin) [S, head_dim] in the interleaved-pair convention of diffusers' apply_rotary_emb."""
    pos = torch.arange(S, dtype=torch.float32)[:, None]
    inv = 1.0 / (10000 ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    ang = pos * inv[None, :]
    return ang.cos().repeat_interleave(2, dim=1), ang.sin().repeat_interleave(2, dim=1)


def wan_freqs(S, head_dim):
    pos = torch.arange(S, dtype=torch.float64)[:, None]
    inv = 1.0 / (10000 ** (torch.arange(0, head_dim, 2, dtype=torch.float64) / head_dim))
    return torch.polar(torch.ones(S, head_dim // 2, dtype=torch.float64), pos * inv[None, :])[None, None]


def wan22_rope(S, head_dim):
    """(freqs_cos, freqs_sin) [1, S, 1, head_dim] as diffusers' Wan2.2 rotary module hands them over: every angle
    repeated for its (even, odd) channel pair; the processor reads cos[..., 0::2] and sin[..., 1::2]."""
    pos = torch.arange(S, dtype=torch.float32)[:, None]
    inv = 1.0 / (10000 ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    ang = (pos * inv[None, :]).repeat_interleave(2, dim=1)
    return ang.cos()[None, :, None, :], ang.sin()[None, :, None, :]


def fake_attn_wan_i2v(seed, heads, head_dim):
    """Wan I2V cross-attention module: text keys + an image context through add_k_proj / add_v_proj / norm_added_k."""
    a = fake_attn(seed, heads, head_dim, wan=True)
    dim = heads * head_dim
    a.add_k_proj, a.add_v_proj = _linear(seed, 6, dim, dim), _linear(seed, 7, dim, dim)
    a.norm_added_k = RMS(dim)
    a.modules_ += [a.add_k_proj, a.add_v_proj, a.norm_added_k]
    return a


# TeaCache pinning cases (tests/golden/make_golden.py::teacache ran the reference's own forwards on these sequences)
TEACACHE_HUNYUAN_CASES = [  # (seed, drift, rel_l1_thresh, num_steps)
    (1, 0.02, 0.15, 20), (2, 0.05, 0.1, 20), (3, 0.2, 0.3, 12)]
TEACACHE_WAN_CASES = [  # (seed, drift, teacache_thresh, steps, model size key of the script, use_ret_steps)
    (7, 0.004, 0.2, 12, "14B", True), (8, 0.05, 0.2, 12, "14B", False), (9, 0.003, 0.1, 10, "1.3B", True),
    (10, 0.02, 0.08, 10, "1.3B", False)]


TEACACHE_COG_CASES = [  # (seed, drift, rel_l1_thresh, num_steps, checkpoint name = key of the script's coefficients_dict)
    (21, 0.03, 0.2, 16, "CogVideoX1.5-5B"), (22, 0.01, 0.1, 16, "CogVideoX-5b"), (23, 0.05, 0.3, 12, "CogVideoX-2b"),
    (24, 0.03, 0.2, 12, "CogVideoX1.5-5B-I2V")]
TEACACHE_FLUX_CASES = [  # (seed, drift, rel_l1_thresh, num_steps)
    (31, 0.02, 0.4, 16), (32, 0.05, 0.8, 16)]
TEACACHE_WAN22_CASES = [  # (seed, drift, teacache_thresh, num_steps, transformer_steps, model size key): two transformers
    (41, 0.004, 0.2, 12, 5, "14B"), (42, 0.003, 0.1, 10, 4, "1.3B")]


def teacache_sequence(seed, n, drift, shape):
    """Deterministic drifting tensor sequence (counter-based generator): x_i = x_{i-1} + drift*(1+0.5 sin i)*noise_i."""
    x = torch.from_numpy(synth.normal(seed, 0, shape).astype(np.float32))
    seq = []
    for i in range(n):
        x = x + float(drift * (1.0 + 0.5 * np.sin(i))) * torch.from_numpy(
            synth.normal(seed, i + 1, shape).astype(np.float32))
        seq.append(x.clone())
    return seq
