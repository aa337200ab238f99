"""Pins the CPU oracle (oracle/) against vectors produced by the reference's own code
(tests/golden/make_golden.py, run in the build container).  CPU only."""
import os

import numpy as np
import pytest

from conftest import (B2_CASES, BIG_CASES, GOLDEN, HEADLINE_CASES, OP_CASES, case_inputs, headline_inputs, load_headline_case,
                      load_op_case, reference_rows)
from oracle import oracle as orc


@pytest.mark.parametrize("name", OP_CASES + B2_CASES)
def test_operator_matches_reference(name):
    meta, gold = load_op_case(name)
    q, k, v, lay, nbr = case_inputs(meta)
    out, parts = orc.rectified_attention(q, k, v, lay, meta["top_k"], meta["p"], nbr, want_parts=True)
    for b in range(meta["B"]):
        for h in range(meta["H"]):
            sel = parts[b * meta["H"] + h]
            # block mask and GAPR mask: bit-exact
            assert np.array_equal(sel["kept"], gold["one_hot"][b, h]), f"{name}: block mask differs (b{b} h{h})"
            assert np.array_equal(sel["unrel"], gold["nogapr"][b, h]), f"{name}: GAPR mask differs (b{b} h{h})"
            # implicit full attention (IPAR probabilities): fp32 round-off only
            np.testing.assert_allclose(sel["probs"], gold["probs"][b, h], rtol=2e-5, atol=1e-6)
    # whole operator: the reference ran its Triton kernel in fp16 (interpreter), the oracle in fp64
    err = np.abs(out - gold["out"])[reference_rows(meta, lay)]
    assert err.max() < 2e-3, f"{name}: max|dO| = {err.max()}"
    assert err.mean() < 2e-4


@pytest.mark.parametrize("name", BIG_CASES)
def test_long_row_case_matches_reference(name):
    """Rows longer than 128 columns (Wan 260 blocks; round 3: HunyuanVideo 136 visual blocks + text tail, Flux 132 + 512
    text tokens -- IPAR and the text columns on K3's sorted-head path): masks / GAPR / probabilities of every row, the
    output on sampled query blocks (the fp64 dense-masked restatement of every row would take minutes on CPU; the GPU
    test compares every row)."""
    meta, gold = load_op_case(name)
    q, k, v, lay, nbr = case_inputs(meta)
    sel = orc.select_head(q[0, 0], k[0, 0], v[0, 0], lay, meta["top_k"], meta["p"], nbr)
    assert np.array_equal(sel["kept"], gold["one_hot"][0, 0]) and np.array_equal(sel["unrel"], gold["nogapr"][0, 0])
    np.testing.assert_allclose(sel["probs"], gold["probs"][0, 0], rtol=2e-5, atol=1e-6)
    rows = sorted({0, 1, 97, lay.NBv // 2, lay.NBv - 2, lay.NBv - 1})
    o = orc.sparse_attention_head(q[0, 0], k[0, 0], v[0, 0], lay, sel["kept"][rows], rows)
    o = o * sel["R"][rows][:, None, None].astype(np.float64) + sel["comp"][rows][:, None, :].astype(np.float64)
    for a, i in enumerate(rows):
        err = np.abs(o[a] - gold["out"][0, i * 128:(i + 1) * 128].astype(np.float64))
        assert err.max() < 2.5e-3, f"{name}: block {i}: {err.max()}"


@pytest.mark.parametrize("name", HEADLINE_CASES)
def test_headline_size_mask_selection_matches_reference(name):
    """BASELINE's FULL sizes (900 + 2 / 512 + 4 / 591 key blocks per row): every row of one head of the oracle's mask
    selection against the reference's own `_build_block_index_with_importance_optimized` run at that size
    (make_golden.py headline: one-hot mask, GAPR mask, num_blocks_needed).  A differing row is reported with its margins,
    not reseeded away."""
    meta, gold = load_headline_case(name)
    assert gold["mismatch_rows"].size == 0, f"{name}: generator recorded rows differing from the oracle: {gold['mismatch_margin']}"
    q, k, v, lay, nbr = headline_inputs(meta)
    if lay.pool_valid < lay.S:
        k, v = k.copy(), v.copy()
        k[lay.pool_valid:] = 0
        v[lay.pool_valid:] = 0
    sel = orc.select_head(q, k, v, lay, meta["top_k"], meta["p"], nbr)
    bad = [i for i in range(lay.NBv) if not np.array_equal(sel["kept"][i], gold["one_hot"][i])]
    assert not bad, f"{name}: kept mask differs from the reference on rows {bad[:8]} ({len(bad)} rows)"
    badg = [i for i in range(lay.NBv) if not np.array_equal(sel["unrel"][i], gold["nogapr"][i])]
    assert not badg, f"{name}: GAPR mask differs from the reference on rows {badg[:8]} ({len(badg)} rows)"
    assert np.array_equal(np.maximum(sel["n_needed"], meta["top_k"]), gold["num_blocks_needed"])
    step = max(1, lay.NBv // 8)
    np.testing.assert_allclose(sel["probs"][::step], gold["probs_sample"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(sel["probs"].sum(-1), gold["probs_rowsum"], rtol=1e-5)


def test_estimate_pr_gain_matches_reference():
    from rectified_spaattn_amd import synth
    z = np.load(os.path.join(GOLDEN, "gapr_1024.npz"))
    shape = tuple(z["shape"])
    gold = np.unpackbits(z["mask"], axis=-1)[..., : shape[-1]].astype(bool)
    q, k, _ = synth.structured_qkv(int(z["seed"]), 1, 2, 1024, 128)
    for h in range(2):
        qbar, aq = orc.pool(q[0, h], 1024, 8, True)
        kbar, ak = orc.pool(k[0, h], 1024, 8, True)
        np.testing.assert_allclose(qbar, z["q_pools"][0, h], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(kbar, z["k_pools"][0, h], rtol=1e-5, atol=1e-6)
        s = orc.dots(qbar, kbar)
        np.testing.assert_allclose(s, z["scores"][0, h], rtol=1e-4, atol=1e-4)
        unrel = ~(np.abs(s) > (np.abs(orc.dots(aq, kbar)) + np.abs(orc.dots(qbar, ak))))
        assert np.array_equal(unrel, gold[0, h])


def test_dense_matches_reference():
    from rectified_spaattn_amd import synth
    z = np.load(os.path.join(GOLDEN, "dense_1536.npz"))
    q, k, v = synth.structured_qkv(int(z["seed"]), 1, 1, 1536, 128)
    n = int(z["n_valid"])
    o = orc.dense_attention(q[0, 0], k[0, 0], v[0, 0])
    np.testing.assert_allclose(o, z["torch"][0, 0], atol=2e-5)
    om = orc.dense_attention(q[0, 0], k[0, 0], v[0, 0], kv_valid=n)
    np.testing.assert_allclose(om, z["vanilla_masked"][0, 0], atol=2e-5)
    # flash mode also returns [b, a, s, d] (attn.py:153): rows < n attend kv < n ; rows >= n attend kv >= n
    fl = z["flash_varlen"][0, 0]
    np.testing.assert_allclose(om[:n], fl[:n], atol=2e-5)
    # causal=True of the reference ("torch" == "vanilla", attn.py:105, :129-133)
    oc = orc.dense_attention(q[0, 0], k[0, 0], v[0, 0], causal=True)
    np.testing.assert_allclose(oc, z["torch_causal"][0, 0], atol=2e-5)


def test_exp_contract_accuracy():
    xs = np.linspace(-80, 0, 4001, dtype=np.float32)
    got = np.array([orc.orc_exp(float(x)) for x in xs])
    ref = np.exp(xs.astype(np.float64))
    rel = np.abs(got - ref) / ref
    assert rel.max() < 2e-5
    assert orc.orc_exp(0.0) == 1.0
    assert orc.orc_exp(-1000.0) == 0.0


def test_keep_all_equals_dense():
    """top_k = NB  =>  R == 1, comp == 0, output == dense attention (SURVEY section 4 invariant)."""
    from rectified_spaattn_amd import synth
    q, k, v = synth.structured_qkv(3, 1, 1, 700, 64)
    lay = orc.layout_wan(700, 0)
    out, parts = orc.rectified_attention(q, k, v, lay, lay.NBv, 0.3, None, want_parts=True)
    assert parts[0]["kept"].all()
    np.testing.assert_allclose(parts[0]["R"], 1.0, atol=1e-5)
    assert np.abs(parts[0]["comp"]).max() == 0
    dense = orc.dense_attention(q[0, 0], k[0, 0], v[0, 0])
    np.testing.assert_allclose(out[0], dense, atol=1e-5)


def test_bit_packing_roundtrip():
    rng = np.random.default_rng(0)
    m = (rng.random((3, 5, 77)) < 0.3).astype(np.uint8)
    assert np.array_equal(orc.unpack_bits(orc.pack_bits(m), 77), m)


def test_no_operator_fixture_was_made_by_rejecting_seeds():
    """make_golden.py tries seeds FIRST_SEED, FIRST_SEED + 1, ... until the oracle and the reference agree on every mask bit.
    Every committed op_* fixture carries the FIRST seed of that sequence: none was produced by skipping a seed on which the two
    disagreed (a fixture regenerated after a rejection stores the rejected seeds with the failing rows' margins in its meta, and
    this test then demands that record)."""
    import ast
    import glob
    import re
    src = open(os.path.join(GOLDEN, "make_golden.py")).read()
    first = int(re.search(r"^FIRST_SEED = (\d+)", src, re.M).group(1))
    files = sorted(glob.glob(os.path.join(GOLDEN, "op_*.npz")))
    assert len(files) >= 17
    for f in files:
        meta = ast.literal_eval(str(np.load(f)["meta"]))
        skipped = meta["seed"] - first
        assert skipped >= 0
        rej = meta.get("rejected_seeds", [])
        assert len(rej) == skipped, f"{os.path.basename(f)}: seed {meta['seed']} but {len(rej)} rejected seeds recorded"
        assert [r["seed"] for r in rej] == list(range(first, meta["seed"]))
        for r in rej:
            assert r["rows"], "a rejected seed must name the rows that differed and their margins"
