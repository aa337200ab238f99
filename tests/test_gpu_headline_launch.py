"""K5 at the SIZE OF THE HEADLINE LAUNCH (round 6; VERDICT r5 "Missing 2").  The bench times one launch over all 24 heads x 902
blocks of the HunyuanVideo 720p shape, while every other full-size parity test runs 2-4 heads -- and the 64-row kernel plans
its work mapping, its aligned starts and its tail split from the size of the launch (rsa_attn_kernel64.hip::k5w_map,
rsa_attn.hip::launch_attn).  Here the launch itself is checked:

  * 24 heads, regime R2 (top_k 90, no neighbours): aligned starts on / off give the same BYTES; sampled query blocks and text
    rows of three heads agree with the oracle (fp64 dense-masked attention over the device's own kept rows, x R + comp) and
    `bench.check_output` -- the check the bench line itself carries -- says ok;
  * 3 heads (one rank of the 8-GPU split; the tail split fires: 2 712 sparse workgroups = 5 generations + 152 blocks): against the
    oracle, and against the same three heads computed inside the 24-head launch (within rounding on the split blocks, byte for byte
    elsewhere).
"""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
S, NT, TOP_K = 115456, 115400, 90


def _call(H, head0=0):
    import bench
    from rectified_spaattn_amd import _core
    wl = bench.WORKLOADS["hunyuan_720p_128f"]
    spec = bench.make_spec(wl)
    q, k, v = bench.gen_inputs(wl, H, head0, torch.device(DEV), "iid")
    return _core.StagedCall(q, k, v, spec, TOP_K, 0.0, None, reuse_buffers=False), spec


def _oracle_blocks(call, lay, h, blocks):
    """max / mean |dO| of query blocks `blocks` and the first / last text row of local head h against the oracle, with the kept
    rows, R and comp the DEVICE produced (the selection pass is compared bit for bit in tests/test_gpu_fullsize.py)."""
    from rectified_spaattn_amd import _core
    qh, kh, vh = (x[0, h].float().cpu().numpy() for x in (call.q, call.k, call.v))
    kh[lay.pool_valid:] = 0
    vh[lay.pool_valid:] = 0
    kept = _core.unpack_bitmask(call.bufs["bitmask"][h:h + 1], lay.NB_total)[0].cpu().numpy()
    R = call.bufs["R"][h].cpu().numpy()
    comp = call.bufs["comp"][h].cpu().numpy()
    ref = orc.sparse_attention_head(qh, kh, vh, lay, kept[blocks].astype(np.uint8), blocks)
    ref = ref * R[blocks][:, None, None] + comp[blocks][:, None, :]
    worst, mean = 0.0, 0.0
    for a, i in enumerate(blocks):
        got = call.out[0, i * 128:(i + 1) * 128, h].float().cpu().numpy()
        err = np.abs(got - ref[a])
        worst, mean = max(worst, float(err.max())), max(mean, float(err.mean()))
    r0 = lay.NBv * 128
    rows = [r0, r0 + lay.q_text_valid - 1]
    reft = orc.dense_attention(qh[rows], kh, vh, lay.kv_text_valid)
    worst = max(worst, float(np.abs(call.out[0, rows, h].float().cpu().numpy() - reft).max()))
    assert float(call.out[0, r0 + lay.q_text_valid:, h].abs().max()) == 0.0, "padded text rows must be 0"
    return worst, mean


def test_the_24_head_headline_launch_is_checked_and_aligned_starts_do_not_change_a_byte():
    import bench
    from rectified_spaattn_amd import _lib
    call, spec = _call(24)
    lay = orc.layout_hunyuan(S, NT)
    L = _lib.lib()
    call.select()
    outs = []
    try:
        for gs in (1, 0, 1):
            assert L.rsa_set_tuning(b"k5_gsync", gs) == 0
            call.attend()
            torch.cuda.synchronize()
            outs.append(call.out.view(torch.int16).clone())
    finally:
        L.rsa_set_tuning(b"k5_gsync", 1)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "aligned starts changed the output bytes"
    counts = call.bufs["counts"]
    assert int(counts.min()) >= TOP_K + 1 and int(counts.max()) == TOP_K + 2      # R2: top_k columns (the text column may be one of them) + the two text blocks
    chk = bench.check_output(call, spec)
    assert chk["ok"] and chk["finite"], chk
    assert chk["blocks"] >= 24 and chk["text_rows"] == 3 * spec.q_text_valid
    for h in (0, 13, 23):
        worst, mean = _oracle_blocks(call, lay, h, [0, 451, 899])
        assert worst <= 2e-2 and mean <= 2e-3, (h, worst, mean)


def test_the_three_heads_of_one_rank_with_the_tail_split_against_the_oracle_and_the_whole_launch():
    from rectified_spaattn_amd import _lib
    lay = orc.layout_hunyuan(S, NT)
    whole, _ = _call(24)
    whole.select(); whole.attend()
    torch.cuda.synchronize()
    ref3 = whole.out[:, :, 21:24].clone()                  # heads 21..23 = rank 7 of 8
    del whole
    torch.cuda.empty_cache()
    part, spec = _call(3, head0=21)
    part.select(); part.attend()
    torch.cuda.synchronize()
    assert torch.isfinite(part.out).all()
    # the tail split fires at this size: 3 x 904 = 2 712 sparse workgroups = 5 generations of 512 + 152 blocks, 96 text pieces behind
    n_sparse = 3 * ((spec.NBv + 7) // 8 * 8)
    assert n_sparse % 512 and n_sparse // 512 >= 1
    diff = (part.out.float() - ref3.float()).abs()
    assert float(diff.max()) <= 2 * 2.0 ** -7 * max(1.0, float(ref3.float().abs().max())), float(diff.max())
    frac_same = float((part.out.view(torch.int16) == ref3.view(torch.int16)).float().mean())
    assert frac_same >= 0.90, frac_same      # the unsplit generations are byte-identical; only tail blocks / text pieces may round differently
    for h in (0, 2):
        # a block of the first generation, one of the middle, and the LAST query blocks of the mapping's tail
        worst, mean = _oracle_blocks(part, lay, h, [0, 450, 898, 899])
        assert worst <= 2e-2 and mean <= 2e-3, (h, worst, mean)
    # and with the split switched off the same three heads are byte-identical to the whole launch's only where the text split agrees;
    # against the oracle they must hold regardless
    L = _lib.lib()
    try:
        assert L.rsa_set_tuning(b"k5_tail_split", 0) == 0
        part.attend()
        torch.cuda.synchronize()
    finally:
        L.rsa_set_tuning(b"k5_tail_split", 1)
    worst, mean = _oracle_blocks(part, lay, 1, [7, 899])
    assert worst <= 2e-2 and mean <= 2e-3, (worst, mean)
