"""oracle/sparse_cpu.py (the bench's like-for-like CPU column) against the oracle's dense-masked fp64 restatement."""
import numpy as np
import torch

from oracle import oracle as orc
from oracle import sparse_cpu
from rectified_spaattn_amd import synth


def test_sparse_cpu_port_matches_oracle_blocks():
    S, D, top_k, p = 10 * 128 + 256, 64, 3, 0.3
    num_true = 10 * 128 + 200
    q, k, v = synth.structured_qkv(77, 1, 1, S, D)
    lay = orc.layout_hunyuan(S, num_true)
    nbr = synth.banded_neighbors(lay.NBv, 1)
    qq, kk, vv = q[0, 0], k[0, 0].copy(), v[0, 0].copy()
    kk[lay.pool_valid:] = 0
    vv[lay.pool_valid:] = 0
    sel = orc.select_head(qq, kk, vv, lay, top_k, p, nbr)
    ref = orc.sparse_attention_head(qq, kk, vv, lay, sel["kept"], sel["rows"])
    ref = ref * sel["R"][:, None, None].astype(np.float64) + sel["comp"][:, None, :].astype(np.float64)
    kept = sel["kept"].astype(bool)
    NBv, NB = kept.shape
    cols = np.zeros((NBv, NB), np.int32)
    counts = kept.sum(1).astype(np.int32)
    for i in range(NBv):
        cols[i, :counts[i]] = np.nonzero(kept[i])[0]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    out = sparse_cpu.rectified_sparse_blocks_cpu(t(qq), t(kk), t(vv), t(cols), t(counts), t(sel["R"]),
                                                 t(sel["comp"]), lay.kv_valid, list(sel["rows"]))
    err = np.abs(out.double().numpy() - ref)
    assert err.max() < 1e-4, err.max()
