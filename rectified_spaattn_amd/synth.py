"""Deterministic synthetic Q/K/V for tests, fixtures and the benchmark.

Counter-based (SplitMix64 -> Box-Muller), so the same (seed, shape) gives the same values on any machine,
numpy version or device -- torch.manual_seed streams are not stable across versions.  Values follow the
structured generator of SURVEY.md section 8(d): every 128-token block shares a centroid, Q and K share the
same centroids (so pooled scores are informative and the selected mask is realistically sparse), V ~ N(0,1).
"""
from __future__ import annotations

import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform(seed: int, stream: int, n: int) -> np.ndarray:
    """n doubles in (0, 1)."""
    with np.errstate(over="ignore"):
        base = _mix(np.uint64(seed) * np.uint64(0x632BE59BD9B4E019) + np.uint64(stream))
        ctr = np.arange(n, dtype=np.uint64) + base
    bits = _mix(ctr) >> np.uint64(11)
    return (bits.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(seed: int, stream: int, shape) -> np.ndarray:
    n = int(np.prod(shape))
    u1 = uniform(seed, 2 * stream, n)
    u2 = uniform(seed, 2 * stream + 1, n)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).reshape(shape)


def round_bf16(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)) << np.uint64(16)
    return (r & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.float32).reshape(np.shape(x))


def structured_qkv(seed: int, B: int, H: int, S: int, D: int, block: int = 128, c: float = 1.5,
                   sigma: float = 0.5, smooth: float = 0.0):
    """Returns q, k, v as fp32 arrays [B, H, S, D] whose values are exactly representable in bf16.

    x[b,h,blk*128+r,:] = c*u[b,h,blk] + sigma*eps ; `smooth` in [0,1) makes consecutive centroids correlated
    (AR(1)), mimicking spatial coherence along the Hilbert curve.
    """
    nb = (S + block - 1) // block
    u = normal(seed, 1, (B, H, nb, D))
    if smooth > 0.0:
        for i in range(1, nb):
            u[:, :, i] = smooth * u[:, :, i - 1] + np.sqrt(1.0 - smooth * smooth) * u[:, :, i]
    cent = np.repeat(u, block, axis=2)[:, :, :S]
    q = c * cent + sigma * normal(seed, 2, (B, H, S, D))
    k = c * cent + sigma * normal(seed, 3, (B, H, S, D))
    v = normal(seed, 4, (B, H, S, D))
    return round_bf16(q), round_bf16(k), round_bf16(v)


def banded_neighbors(nb: int, width: int = 1) -> np.ndarray:
    """Stand-in block-neighbour matrix: |i-j| <= width (symmetric, True diagonal)."""
    i = np.arange(nb)
    return np.abs(i[:, None] - i[None, :]) <= width
