"""Gilbert-curve geometry: the library's C++ enumerator vs vectors produced by the reference's
utils/jenga_gilbert.py (tests/golden/gilbert.npz).  Host code: runs without a GPU."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN
from rectified_spaattn_amd.utils import jenga_gilbert as jg

G = np.load(os.path.join(GOLDEN, "gilbert.npz"))
SHAPES = [(4, 12, 16), (1, 32, 32), (2, 6, 10), (3, 5, 7), (5, 4, 3)]
ORDERS = [("w", "h", "t"), None, ("t", "h", "w")]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("ao", ORDERS)
def test_mapping_and_neighbors_match_reference(shape, ao):
    t, h, w = shape
    tag = f"{t}x{h}x{w}_{''.join(ao) if ao else 'auto'}"
    l2h, h2l = jg.gilbert_mapping(t, h, w, axis_order=ao)
    assert np.array_equal(np.asarray(l2h, np.int32), G[f"l2h_{tag}"])
    assert np.array_equal(np.asarray(h2l, np.int32), G[f"h2l_{tag}"])
    nb = jg.gilbert_block_neighbor_mapping(t, h, w, block_size=16, axis_order=ao)
    n = int(G[f"nbr16n_{tag}"])
    gold = np.unpackbits(G[f"nbr16_{tag}"], axis=-1)[:, :n].astype(bool)
    assert nb.shape == (n, n) and np.array_equal(nb.numpy(), gold)
    # invariants: permutation + inverse, symmetric relation with a True diagonal
    assert sorted(l2h) == list(range(t * h * w))
    assert all(h2l[l2h[i]] == i for i in range(0, t * h * w, 7))
    assert bool((nb == nb.T).all()) and bool(nb.diagonal().all())


def test_hunyuan_latent_full_size_digest():
    l2h, h2l = jg.gilbert_mapping(32, 45, 80, axis_order=("w", "h", "t"))
    assert hashlib.sha256(np.asarray(l2h, np.int32).tobytes()).hexdigest() == str(G["hunyuan_l2h_sha256"])
    nb = jg.gilbert_block_neighbor_mapping(32, 45, 80, axis_order=("w", "h", "t"))
    assert nb.shape == (900, 900)
    assert hashlib.sha256(nb.numpy().astype(np.uint8).tobytes()).hexdigest() == str(G["hunyuan_nbr_sha256"])
    assert np.array_equal(nb.sum(1).numpy().astype(np.int32), G["hunyuan_nbr_rowsum"])


def test_argument_errors():
    with pytest.raises(ValueError):
        jg.gilbert_mapping(2, 2, 2, transpose_order=[2, 1])
    with pytest.raises(ValueError):
        jg.gilbert_mapping(2, 2, 2, axis_order=("w", "w", "t"))


def test_transpose_order_matches_reference():
    """gilbert_mapping(transpose_order=...) (reference utils/jenga_gilbert.py:290-346, :458-504): the curve of the permuted box
    read at the permuted coordinates; gilbert_block_neighbor_mapping accepts the argument and ignores it, as the reference does."""
    import torch
    for (t, h, w) in [(2, 6, 10), (3, 5, 7), (4, 12, 16)]:
        for to in ([2, 1, 0], [1, 0, 2], [0, 2, 1]):
            tag = f"{t}x{h}x{w}_{''.join(map(str, to))}"
            l2h, h2l = jg.gilbert_mapping(t, h, w, transpose_order=to)
            assert np.array_equal(np.asarray(l2h, np.int32), G[f"tr_l2h_{tag}"]), tag
            assert np.array_equal(np.asarray(h2l, np.int32), G[f"tr_h2l_{tag}"]), tag
            assert torch.equal(jg.gilbert_block_neighbor_mapping(t, h, w, block_size=16, transpose_order=to),
                               jg.gilbert_block_neighbor_mapping(t, h, w, block_size=16))
    import pytest
    with pytest.raises(ValueError):
        jg.gilbert_mapping(2, 3, 4, transpose_order=[0, 1, 1])
