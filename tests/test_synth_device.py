"""The device-side input generator (rectified_spaattn_amd/synth_device.py) is the same counter-based stream as
synth.py (run here on CPU torch tensors): uniforms bit-equal, normals to fp64 rounding, bf16 inputs equal."""
import numpy as np
import torch

from rectified_spaattn_amd import synth, synth_device


def test_uniform_and_normal_streams_match_numpy_generator():
    for seed, stream in ((7, 3), (20251212, 1), (0, 0)):
        u = synth_device.uniform(seed, stream, 4096, "cpu").numpy()
        assert np.array_equal(u, synth.uniform(seed, stream, 4096))
    n = synth_device.normal(11, 2, (257, 16), "cpu").numpy()
    assert np.abs(n - synth.normal(11, 2, (257, 16))).max() < 1e-14


def test_structured_qkv_heads_equal_per_head_numpy_inputs():
    S, D = 700, 64
    q, k, v = synth_device.structured_qkv_device(100, 2, 5, S, D, "cpu")
    for h in range(2):
        q0, k0, v0 = synth.structured_qkv(105 + h, 1, 1, S, D)
        for got, ref in ((q, q0), (k, k0), (v, v0)):
            assert (got[0, h].float().numpy() == ref[0, 0]).mean() > 0.9999


def test_head_sharding_generates_the_same_global_tensor():
    full = synth_device.structured_qkv_device(9, 4, 0, 300, 64, "cpu")
    part = synth_device.structured_qkv_device(9, 2, 2, 300, 64, "cpu")
    for f, p in zip(full, part):
        assert torch.equal(f[:, 2:], p)
