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
        return _transposed_mapping((int(t), int(h), int(w)), transpose_order)
    n = int(t) * int(h) * int(w)
    l2h = np.empty(n, np.int32)
    h2l = np.empty(n, np.int32)
    _lib.check(_lib.lib().rsa_gilbert_mapping(int(t), int(h), int(w), _axis(axis_order),
                                              l2h.ctypes.data_as(ctypes.c_void_p),
                                              h2l.ctypes.data_as(ctypes.c_void_p)), "rsa_gilbert_mapping")
    return l2h.tolist(), h2l.tolist()


def _transposed_mapping(dims, order):
    """reference transpose_gilbert_mapping (:290-346): the curve of the box (T, H, W) = dims[order] (built with the reference's
    default axis choice, axis_order=None), read at the permuted coordinates: the point c of the ORIGINAL box (linear index
    row-major over dims) sits at (z, y, x) = (c[order[0]], c[order[1]], c[order[2]]) of the transposed one."""
    order = [int(o) for o in order]
    if len(order) != 3 or set(order) != {0, 1, 2}:
        raise ValueError("order must be a permutation of 0,1,2")
    T, H, W = (dims[o] for o in order)
    n = T * H * W
    base, inv = np.empty(n, np.int32), np.empty(n, np.int32)
    _lib.check(_lib.lib().rsa_gilbert_mapping(T, H, W, None, base.ctypes.data_as(ctypes.c_void_p),
                                              inv.ctypes.data_as(ctypes.c_void_p)), "rsa_gilbert_mapping")
    c = np.indices(dims).reshape(3, -1)                       # coordinates of every original point, row-major order
    l2h = base[(c[order[0]] * H + c[order[1]]) * W + c[order[2]]]
    h2l = np.empty(n, np.int32)
    h2l[l2h] = np.arange(n, dtype=np.int32)
    return l2h.tolist(), h2l.tolist()


def gilbert_block_neighbor_mapping(t, h, w, block_size=128, transpose_order=None, axis_order=("w", "h", "t")):
    """bool tensor [NB, NB]: block i and block j contain 26-neighbouring points (diagonal True).  `transpose_order` is accepted
    and has no effect, as in the reference (its body :613-693 never reads the argument)."""
    n = int(t) * int(h) * int(w)
    nb = (n + block_size - 1) // block_size
    out = np.empty((nb, nb), np.uint8)
    _lib.check(_lib.lib().rsa_gilbert_block_neighbors(int(t), int(h), int(w), int(block_size), _axis(axis_order),
                                                      out.ctypes.data_as(ctypes.c_void_p)),
               "rsa_gilbert_block_neighbors")
    return torch.from_numpy(out).bool()
