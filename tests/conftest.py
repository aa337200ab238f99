import ast
import os
import sys

import numpy as np
import pytest

os.environ.setdefault("RSA_TUNING", "1")  # the variant tests flip kernel forms through rsa_set_tuning (debug hook)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_op_case(name):
    """Load tests/golden/op_<name>.npz -> (meta dict, arrays dict with masks unpacked)."""
    z = np.load(os.path.join(GOLDEN, f"op_{name}.npz"))
    meta = ast.literal_eval(str(z["meta"]))
    oh_shape = tuple(z["one_hot_shape"])
    ng_shape = tuple(z["nogapr_shape"])
    one_hot = np.unpackbits(z["one_hot"], axis=-1)[..., : oh_shape[-1]]
    nogapr = np.unpackbits(z["nogapr"], axis=-1)[..., : ng_shape[-1]]
    return meta, dict(one_hot=one_hot, nogapr=nogapr, probs=z["probs"], out=z["out"])


OP_CASES = ["wan_640", "wan_pad_1450", "wan_d64_1100", "wan_nonbr_1024", "hunyuan_1280", "hunyuan_3328",
            "hunyuan_full_1536", "flux_1536", "cogvideo_994", "wan_smooth_2048"]


# batch of two per layout (round 2).  wan has none: the reference operator cannot run B > 1 (rectified_wan21_attn.py:329
# indexes with a per-batch tensor); tests/test_gpu_parity.py covers B = 2 for wan against the oracle alone.
B2_CASES = ["b2_hunyuan_1280", "b2_flux_1280", "b2_cogvideo_994"]


# a row longer than 256 columns through the reference (exercises K3's sorted-head path); output stored as fp16
BIG_CASES = ["big_wan_33280", "big_hunyuan_17664", "big_flux_17408", "big_cogvideo_17634"]


def reference_rows(meta, lay):
    """bool [B, S]: rows of a golden `out` that the reference computes for that batch item.  B = 1: all.  B = 2: the
    visual rows of both items, the text rows of item 0 only -- the reference's text-row flash call gets the 3-entry
    cu_seqlens its callers build (rectified_hunyuan_attn.py:371-380), which describes ONE batch item of the packed
    [(b s), a, d] tensor; item 1's text rows are checked against the oracle instead."""
    ok = np.ones((meta["B"], meta["S"]), bool)
    if meta["B"] > 1:
        ok[1:, lay.NBv * 128:] = False
    return ok


def case_inputs(meta):
    """Regenerate the fixture's inputs (counter-based PRNG) and its oracle Layout + neighbour matrix."""
    from oracle import oracle as orc
    from rectified_spaattn_amd import synth
    q, k, v = synth.structured_qkv(meta["seed"], meta["B"], meta["H"], meta["S"], meta["D"],
                                   smooth=meta.get("smooth", 0.0))
    var = meta["variant"]
    if var == "hunyuan":
        lay = orc.layout_hunyuan(meta["S"], meta["num_true"])
    elif var == "flux":
        lay = orc.layout_flux(meta["S"], meta["text_length"])
    elif var == "cogvideo":
        lay = orc.layout_cogvideo(meta["S"], meta["text_length"])
    else:
        lay = orc.layout_wan(meta["S"], meta.get("ffb", 0))
    nbr = synth.banded_neighbors(lay.NBv, meta["nb_width"]) if meta["nb_width"] >= 0 else None
    return q, k, v, lay, nbr


# round 4: the reference's own mask builder at BASELINE's FULL sizes, one head (make_golden.py headline)
HEADLINE_CASES = ["hunyuan_115456", "flux_66048", "wan_75600"]


def load_headline_case(name):
    """tests/golden/headline_<name>.npz -> (meta, dict one_hot [NBv, NB_total], nogapr [NBv, NBv], num_blocks_needed [NBv],
    probs_rowsum, probs_sample (every NBv//8-th row), mismatch_rows / mismatch_margin as recorded by the generator)."""
    z = np.load(os.path.join(GOLDEN, f"headline_{name}.npz"))
    meta = ast.literal_eval(str(z["meta"]))
    oh = np.unpackbits(z["one_hot"], axis=-1)[..., : int(z["one_hot_shape"][-1])]
    ng = np.unpackbits(z["nogapr"], axis=-1)[..., : int(z["nogapr_shape"][-1])]
    return meta, dict(one_hot=oh, nogapr=ng, num_blocks_needed=z["num_blocks_needed"].astype(np.int32),
                      probs_rowsum=z["probs_rowsum"], probs_sample=z["probs_sample"],
                      mismatch_rows=z["mismatch_rows"], mismatch_margin=z["mismatch_margin"])


def headline_inputs(meta):
    """One head of the headline fixture: q, k, v [S, D] fp32 (bf16-representable; the numpy twin of the device generator),
    oracle Layout, neighbour matrix (true Gilbert neighbours for the HunyuanVideo case).  k / v rows past the layout's
    pooling limit are zeroed, as the reference operator does in place (hunyuan :307-308)."""
    from oracle import oracle as orc
    from rectified_spaattn_amd import synth
    from rectified_spaattn_amd.utils import jenga_gilbert
    S, D = meta["S"], meta["D"]
    q, k, v = synth.structured_qkv(meta["seed"], 1, 1, S, D)
    var = meta["variant"]
    nbr = None
    if var == "hunyuan":
        lay = orc.layout_hunyuan(S, meta["num_true"])
        nbr = jenga_gilbert.gilbert_block_neighbor_mapping(*meta["gilbert"]).numpy()
    elif var == "flux":
        lay = orc.layout_flux(S, meta["text_length"])
    else:
        lay = orc.layout_wan(S, meta["ffb"])
    return q[0, 0], k[0, 0], v[0, 0], lay, nbr
