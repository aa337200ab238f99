"""CPU restatement of the RECTIFIED SPARSE path for a given block mask, used as bench.py's like-for-like CPU column
(`cpu_baseline.sparse`, kind "port") -- SURVEY.md 8(d) "also time the CPU restatement of the rectified sparse path
for the same mask".

TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/oracle.py header).  The reference's sparse kernel is Triton and
has no CPU execution mode besides the (un-timeable) interpreter, so this is a port: per 128-row query block, exact
attention over the keys of its kept blocks (semantics of rectified_hunyuan_attn.py:15-105: kept blocks only, kv
columns >= kv_valid masked), then `out * R + comp` (hunyuan :352-365), on the host's cores with PyTorch CPU ops in
the input dtype.  tests/test_sparse_cpu.py checks it against oracle.sparse_attention_head."""
import torch
import torch.nn.functional as F

BLOCK = 128


def rectified_sparse_blocks_cpu(q, k, v, cols, counts, R, comp, kv_valid, qblocks):
    """q, k, v: [S, D] CPU tensors of one head (bf16 / fp16 / fp32); cols [NBv, NB] int (ascending kept block ids,
    first counts[i] valid), counts [NBv], R [NBv] fp32, comp [NBv, D] fp32; qblocks: iterable of q-block ids.
    Returns [len(qblocks), 128, D] in q.dtype (rows past S are computed on zero padding)."""
    S, D = k.shape
    nbp = (S + BLOCK - 1) // BLOCK
    if nbp * BLOCK != S:
        pad = nbp * BLOCK - S
        k = torch.cat([k, k.new_zeros(pad, D)], 0)
        v = torch.cat([v, v.new_zeros(pad, D)], 0)
        q = torch.cat([q, q.new_zeros(pad, D)], 0)
    kb, vb = k.view(nbp, BLOCK, D), v.view(nbp, BLOCK, D)
    qblocks = [int(i) for i in qblocks]
    out = q.new_empty(len(qblocks), BLOCK, D)
    # group query blocks by (kept count, valid keys) so that each group is one SDPA call without a mask: lists are
    # ascending, so only the LAST kept block of a row can cross kv_valid and the valid keys are a prefix of the gather
    by_n = {}
    for pos, i in enumerate(qblocks):
        n = int(counts[i])
        last = int(cols[i, n - 1]) if n > 0 else 0
        valid = max(0, (n - 1) * BLOCK + min(BLOCK, int(kv_valid) - last * BLOCK)) if n > 0 else 0
        by_n.setdefault((n, valid), []).append((pos, i))
    for (n, valid), items in by_n.items():
        pos = torch.tensor([p for p, _ in items])
        ids = torch.tensor([i for _, i in items])
        if valid <= 0:
            out[pos] = comp[ids].float()[:, None, :].expand(-1, BLOCK, -1).to(q.dtype)
            continue
        blk = cols[ids, :n].long()                                    # [g, n]
        kg = kb[blk].reshape(len(items), n * BLOCK, D)[:, :valid]
        vg = vb[blk].reshape(len(items), n * BLOCK, D)[:, :valid]
        qg = q.view(nbp, BLOCK, D)[ids]
        o = F.scaled_dot_product_attention(qg[None], kg[None], vg[None])[0]   # [1, g, 128, D]: groups as "heads"
        o = o.float() * R[ids].float()[:, None, None] + comp[ids].float()[:, None, :]
        out[pos] = o.to(q.dtype)
    return out
