"""Device-side twin of synth.py: the same counter-based generator (SplitMix64 -> Box-Muller) written with torch
integer / fp64 element-wise ops, so the benchmark inputs are generated where they are used (per head, on the
rank's own GPU: no host transfer, no dependence on torch.manual_seed streams) -- SURVEY.md 8(d).

head h of `structured_qkv_device(seed, ...)` equals `synth.structured_qkv(seed + h, 1, 1, S, D)` up to the last
ulp of the fp64 log / cos (tests/test_synth_device.py), i.e. the bf16 values agree except for rare rounding ties.

Two centroid models:
  * independent block centroids (regime R2/R1): every 128-token block draws its own N(0, I) centroid;
  * `spatial_centroids`: a smooth random field over the (t, h, w) latent sampled at the blocks' centres along the
    Gilbert curve -- neighbouring blocks (in space, hence mostly along the curve too) look alike, as in real video
    attention maps; this is the "locality" regime of bench.py.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import synth


def _s64(x: int) -> int:
    """uint64 constant -> the int64 with the same bits."""
    x &= 0xFFFFFFFFFFFFFFFF
    return x - (1 << 64) if x >= (1 << 63) else x


_GOLD = _s64(0x9E3779B97F4A7C15)
_M1 = _s64(0xBF58476D1CE4E5B9)
_M2 = _s64(0x94D049BB133111EB)


def _lsr(z: torch.Tensor, k: int) -> torch.Tensor:
    """logical shift right of int64 bit patterns"""
    return (z >> k) & ((1 << (64 - k)) - 1)


def _mix(z: torch.Tensor) -> torch.Tensor:
    z = z + _GOLD
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    return z ^ _lsr(z, 31)


def _stream_base(seed: int, stream: int) -> int:
    """synth.uniform's `base` as a python int (mod 2^64)."""
    z = (seed * 0x632BE59BD9B4E019 + stream) & 0xFFFFFFFFFFFFFFFF
    z = (z + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def uniform(seed: int, stream: int, n: int, device) -> torch.Tensor:
    """n doubles in (0, 1): synth.uniform on `device`."""
    ctr = torch.arange(n, dtype=torch.int64, device=device) + _s64(_stream_base(seed, stream))
    bits = _lsr(_mix(ctr), 11)
    return (bits.to(torch.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(seed: int, stream: int, shape, device) -> torch.Tensor:
    """synth.normal on `device` (fp64)."""
    n = int(np.prod(shape))
    u1 = uniform(seed, 2 * stream, n, device)
    u2 = uniform(seed, 2 * stream + 1, n, device)
    return (torch.sqrt(-2.0 * torch.log(u1)) * torch.cos((2.0 * math.pi) * u2)).reshape(shape)


class SpatialField:
    """Smooth random field over the (t, h, w) latent sampled at the centres of the 128-token blocks of the Gilbert
    order: centroids(seed) -> [nb, D] fp64 with unit variance per component = a Gaussian-kernel average (length
    `corr_len` latent cells) of lattice vectors N(0, I).  Blocks past the visual range (text tail) get independent
    N(0, I) centroids.  The geometry (curve, block centres, kernel weights) is computed once and shared by all heads."""

    def __init__(self, latent, nb: int, corr_len: float = 6.0, block: int = 128, axis_order=("w", "h", "t")):
        from .utils import jenga_gilbert
        T, Hh, W = latent
        _, h2l = jenga_gilbert.gilbert_mapping(T, Hh, W, axis_order=axis_order)
        lin = np.asarray(h2l, dtype=np.int64)
        n_vis = lin.size
        self.nb, self.nbv = nb, (n_vis + block - 1) // block
        z, y, x = lin // (Hh * W), (lin // W) % Hh, lin % W
        pos = np.stack([z, y, x], axis=1).astype(np.float64)
        pad = self.nbv * block - n_vis
        if pad:
            pos = np.concatenate([pos, np.repeat(pos[-1:], pad, axis=0)], axis=0)
        self.centre = pos.reshape(self.nbv, block, 3).mean(axis=1)             # [nbv, 3]
        step = max(1.0, corr_len / 2.0)
        gz, gy, gx = (np.arange(-step, n + step, step) for n in (T, Hh, W))
        nodes = np.stack(np.meshgrid(gz, gy, gx, indexing="ij"), axis=-1).reshape(-1, 3)
        d2 = ((self.centre[:, None, :] - nodes[None, :, :]) ** 2).sum(-1)
        wgt = np.exp(-d2 / (2.0 * corr_len * corr_len))
        self.wgt = wgt / np.sqrt((wgt * wgt).sum(axis=1, keepdims=True))      # unit variance per component
        self.n_nodes = nodes.shape[0]

    def centroids(self, seed: int, D: int) -> np.ndarray:
        cent = np.empty((self.nb, D), np.float64)
        cent[:self.nbv] = self.wgt @ synth.normal(seed, 1, (self.n_nodes, D))
        if self.nb > self.nbv:
            cent[self.nbv:] = synth.normal(seed, 5, (self.nb - self.nbv, D))
        return cent


def spatial_centroids(seed: int, latent, nb: int, D: int, corr_len: float = 6.0, block: int = 128,
                      axis_order=("w", "h", "t")) -> np.ndarray:
    return SpatialField(latent, nb, corr_len, block, axis_order).centroids(seed, D)


def structured_qkv_device(seed: int, H_local: int, head0: int, S: int, D: int, device, dtype=torch.bfloat16,
                          c: float = 1.5, sigma: float = 0.5, block: int = 128, centroid_fn=None):
    """q, k, v [1, H_local, S, D] on `device`; head hl uses seed + head0 + hl (so any head sharding generates the
    same global tensor).  centroid_fn(head_seed) -> [nb, D] numpy centroids; default: independent N(0, I)."""
    q = torch.empty(1, H_local, S, D, dtype=dtype, device=device)
    k = torch.empty_like(q)
    v = torch.empty_like(q)
    nb = (S + block - 1) // block
    for hl in range(H_local):
        hs = seed + head0 + hl
        if centroid_fn is None:
            u = normal(hs, 1, (nb, D), device)
        else:
            u = torch.from_numpy(np.ascontiguousarray(centroid_fn(hs), dtype=np.float64)).to(device)
        cent = (c * u).repeat_interleave(block, dim=0)[:S]
        q[0, hl] = (cent + sigma * normal(hs, 2, (S, D), device)).to(torch.float32).to(dtype)
        k[0, hl] = (cent + sigma * normal(hs, 3, (S, D), device)).to(torch.float32).to(dtype)
        v[0, hl] = normal(hs, 4, (S, D), device).to(torch.float32).to(dtype)
        del cent
    return q, k, v
