#!/usr/bin/env python3
"""A miniature HunyuanVideo-style denoising loop assembled from this package's drop-in pieces (random weights, no
diffusers, no checkpoints) -- what scripts/main_hunyuan.py wires together in the reference:

    Gilbert curve + block-neighbour matrix  ->  token permutation in / out  ->  attention mask from the text lengths
    ->  N dual-stream blocks whose attention runs through RectifiedHunyuanVideoSpaAttnProcessor2_0 (first layer dense,
    the rest rectified-sparse; fused RMSNorm + RoPE producer)  ->  TeaCache step skipping around the block stack.

    python examples/pipeline_demo.py [--steps 8] [--fp8] [--no-teacache]

Prints the per-step TeaCache decisions and the relative L1 distance between the run with sparse layers and an all-dense
run of the same model.  tests/test_gpu_pipeline_demo.py runs it at this size.
"""
import argparse
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import rectified_spaattn_amd as rsa  # noqa: E402
from rectified_spaattn_amd import glue  # noqa: E402
from rectified_spaattn_amd.rectified_hunyuan_attn import RectifiedHunyuanVideoSpaAttnProcessor2_0  # noqa: E402
from rectified_spaattn_amd.teacache import TeaCache  # noqa: E402
from rectified_spaattn_amd.utils import jenga_gilbert  # noqa: E402


class RMSNorm(nn.Module):
    """Same attributes and arithmetic as diffusers' RMSNorm (the processors detect it by name + eps)."""

    def __init__(self, d, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.eps = eps
        self.bias = None

    def forward(self, x):
        v = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        return (x * torch.rsqrt(v + self.eps)).to(self.weight.dtype) * self.weight


class MiniAttention(nn.Module):
    """The attributes a diffusers `Attention` exposes to its processor (dual-stream flavour)."""

    def __init__(self, heads, head_dim):
        super().__init__()
        dim = heads * head_dim
        self.heads = heads
        self.to_q, self.to_k, self.to_v = (nn.Linear(dim, dim) for _ in range(3))
        self.add_q_proj, self.add_k_proj, self.add_v_proj = (nn.Linear(dim, dim) for _ in range(3))
        self.norm_q, self.norm_k, self.norm_added_q, self.norm_added_k = (RMSNorm(head_dim) for _ in range(4))
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Identity()])
        self.to_add_out = nn.Linear(dim, dim)
        self.processor = None

    def forward(self, hidden, enc, mask, rope):
        return self.processor(self, hidden, enc, mask, rope)


class MiniBlock(nn.Module):
    def __init__(self, heads, head_dim):
        super().__init__()
        dim = heads * head_dim
        self.norm1, self.norm1_ctx = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.attn = MiniAttention(heads, head_dim)
        self.ff = nn.Sequential(nn.Linear(dim, 2 * dim), nn.GELU(), nn.Linear(2 * dim, dim))

    def forward(self, hidden, enc, temb, mask, rope):
        a, e = self.attn(self.norm1(hidden) * (1 + temb), self.norm1_ctx(enc), mask, rope)
        hidden = hidden + 0.5 * a
        enc = enc + 0.5 * e
        return hidden + 0.5 * self.ff(hidden), enc


def rope_tables(order, t, h, w, head_dim, device):
    """(cos, sin) [S, head_dim] for the PERMUTED token order (interleaved-pair convention of apply_rotary_emb)."""
    idx = order.to(torch.float32)
    pos = torch.stack([idx // (h * w), (idx // w) % h, idx % w], 1)                       # (t, y, x) of each token
    dims = [head_dim // 4, 3 * head_dim // 8, 3 * head_dim // 8]
    ang = []
    for a, d in enumerate(dims):
        inv = 1.0 / (10000 ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
        ang.append(pos[:, a: a + 1] * inv[None, :])
    ang = torch.cat(ang, 1)
    return ang.cos().repeat_interleave(2, 1).to(device), ang.sin().repeat_interleave(2, 1).to(device)


def run(steps=8, fp8=False, teacache=True, sparse=True, latent=(4, 16, 16), heads=2, head_dim=128, layers=3,
        n_text=180, seed=0, device="cuda:0", verbose=False):
    t, h, w = latent
    dim, S_vis, dt = heads * head_dim, t * h * w, torch.bfloat16
    torch.manual_seed(seed)
    blocks = nn.ModuleList([MiniBlock(heads, head_dim) for _ in range(layers)]).to(device, dt)
    # geometry: Gilbert order, neighbour matrix, permuted RoPE tables, mask
    l2h, h2l = jenga_gilbert.gilbert_mapping(t, h, w)
    l2h = torch.as_tensor(l2h, dtype=torch.int64)
    h2l = torch.as_tensor(h2l, dtype=torch.int64)
    nbr = jenga_gilbert.gilbert_block_neighbor_mapping(t, h, w)
    rope = rope_tables(h2l, t, h, w, head_dim, device)
    enc_mask = torch.zeros(1, 256, dtype=torch.bool, device=device)
    enc_mask[:, :n_text] = True
    mask, _ = glue.build_attention_mask(S_vis, enc_mask)
    top_k = max(1, int(0.25 * (S_vis // 128)))
    for i, b in enumerate(blocks):   # first layer dense (the scripts keep early layers exact), the rest sparse
        mode = "sparse" if (sparse and i > 0) else "flash"
        b.attn.processor = RectifiedHunyuanVideoSpaAttnProcessor2_0(mode, top_k, nbr, 0.3, i)
    old = (rsa.set_qkv_fp8(fp8), rsa.set_dense_fp8(fp8))
    tc = TeaCache.hunyuan(steps, 0.15) if teacache else None
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(1, S_vis, dim, generator=g).to(device, dt)          # latent tokens, linear order
    enc0 = torch.randn(1, 256, dim, generator=g).to(device, dt)
    decisions = []
    try:
        with torch.no_grad():
            for step in range(steps):
                temb = torch.full((1, 1, dim), 0.05 * (steps - step) / steps, device=device, dtype=dt)
                hidden = glue.permute_tokens(x, h2l)                      # hidden_states[:, hilbert_order]
                enc = enc0
                modulated = blocks[0].norm1(hidden) * (1 + temb)
                if tc is None or tc.should_compute(modulated):
                    h_in = hidden.clone()
                    for b in blocks:
                        hidden, enc = b(hidden, enc, temb, mask, rope)
                    if tc is not None:
                        tc.store_residual(hidden, h_in)
                    decisions.append(True)
                else:
                    hidden = tc.apply_residual(hidden)
                    decisions.append(False)
                out = glue.permute_tokens(hidden, l2h)                    # back to linear order
                x = x - 0.1 * (out - x) / steps                          # a stand-in scheduler step
                if verbose:
                    print(f"step {step}: {'compute' if decisions[-1] else 'skip   '}  |x| {x.float().abs().mean():.4f}")
    finally:
        rsa.set_qkv_fp8(old[0])
        rsa.set_dense_fp8(old[1])
    return x, decisions


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--no-teacache", action="store_true")
    args = ap.parse_args()
    x, dec = run(args.steps, args.fp8, not args.no_teacache, verbose=True)
    ref, _ = run(args.steps, False, False, sparse=False)
    rel = ((x.float() - ref.float()).abs().mean() / ref.float().abs().mean()).item()
    print(f"computed {sum(dec)} of {len(dec)} steps; relative L1 vs the all-dense, no-skip, bf16 run: {rel:.3e}")


if __name__ == "__main__":
    main()
