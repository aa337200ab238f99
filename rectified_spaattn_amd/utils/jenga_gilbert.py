"""Gilbert-curve token order and block-neighbour matrix -- the reference's utils/jenga_gilbert.py entry points
(gilbert_mapping :458-504, gilbert_block_neighbor_mapping :613-693) on the library's C++ enumerator
(csrc/rsa_geometry.cpp).  Host-side, run once at start-up."""
import ctypes

import numpy as np
import torch

from .. import _lib


def _axis(axis_order):
    if axis_order is None:
        return None
    s = "".join(axis_order)
    if sorted(s) != ["h", "t", "w"]:
        raise ValueError(f"axis_order must be a permutation of ('w','h','t'), got {axis_order!r}")
    return s.encode()


def gilbert_mapping(t, h, w, transpose_order=None, axis_order=("w", "h", "t")):
    """Returns (linear_to_hilbert, hilbert_to_linear) as python lists of length t*h*w, linear index =
    z*h*w + y*w + x -- same values as the reference."""
    if transpose_order is not None:
        raise NotImplementedError("transpose_order is not used by any reference script and is not implemented")
    n = int(t) * int(h) * int(w)
    l2h = np.empty(n, np.int32)
    h2l = np.empty(n, np.int32)
    _lib.check(_lib.lib().rsa_gilbert_mapping(int(t), int(h), int(w), _axis(axis_order),
                                              l2h.ctypes.data_as(ctypes.c_void_p),
                                              h2l.ctypes.data_as(ctypes.c_void_p)), "rsa_gilbert_mapping")
    return l2h.tolist(), h2l.tolist()


def gilbert_block_neighbor_mapping(t, h, w, block_size=128, transpose_order=None, axis_order=("w", "h", "t")):
    """bool tensor [NB, NB]: block i and block j contain 26-neighbouring points (diagonal True)."""
    if transpose_order is not None:
        raise NotImplementedError("transpose_order is not used by any reference script and is not implemented")
    n = int(t) * int(h) * int(w)
    nb = (n + block_size - 1) // block_size
    out = np.empty((nb, nb), np.uint8)
    _lib.check(_lib.lib().rsa_gilbert_block_neighbors(int(t), int(h), int(w), int(block_size), _axis(axis_order),
                                                      out.ctypes.data_as(ctypes.c_void_p)),
               "rsa_gilbert_block_neighbors")
    return torch.from_numpy(out).bool()
