"""Where does the all-e4m3 K5 sit on the RAGGED last query block of Wan2.2-TI2V (S = 27 280 = 213 x 128 + 16)?

`bench.py --workload wan22_ti2v_720p_121f --qkv-fp8 1` reported max|dO| 0.38 at (head 0, block 213) against the UN-quantised
dense-masked reference (profiles/r06_bench_wan22_fp8.json) while every other block stays below 0.11.  This script separates the
kernel from the number format on that block:

  * device O vs the un-quantised reference, per query block of three heads (the bench's comparison, every block);
  * device O vs the oracle on the DEQUANTISED e4m3 operands with P kept exact (the format's distance) and with P formed as the
    kernel forms it (code map + deferred reference: what is left is fp32 accumulation and codes on a rounding boundary);
  * the oracle on the dequantised operands vs the oracle on the un-quantised ones (no kernel involved: the format alone);
  * the operand images of that head against oracle.fp8_operands, byte for byte.

Run on the GPU box:  python tests/diag/diag_fp8_ragged_block.py [workload]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from rectified_spaattn_amd import _core  # noqa: E402


def per_block_errors(call, spec, h):
    q, k, v, out = call.q, call.k, call.v, call.out
    S, D = q.shape[2], q.shape[3]
    cols, counts, Rb, compb = (call.bufs[n] for n in ("cols", "counts", "R", "comp"))
    ar = torch.arange(128, device=q.device)
    qf, kf, vf = (x[0, h].float() for x in (q, k, v))
    res = []
    for i in range(spec.NBv):
        n = int(counts[h, i].item())
        sel = cols[h, i, :n].long()
        key_idx = (sel[:, None] * 128 + ar[None]).reshape(-1)
        valid = key_idx < spec.kv_valid
        key_idx = key_idx.clamp(max=S - 1)
        rows = slice(i * 128, min(S, (i + 1) * 128))
        sc = (qf[rows] @ kf[key_idx].t()) * float(D) ** -0.5
        sc = sc.masked_fill(~valid[None, :], float("-inf"))
        pr = torch.softmax(sc, dim=-1)
        ref = pr @ vf[key_idx] * Rb[h, i] + compb[h, i][None, :]
        err = (out[0, rows, h].float() - ref).abs()
        res.append((float(err.max()), float(err.mean()), float(pr.max(dim=-1).values.mean()), float(Rb[h, i])))
    return np.array(res)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "wan22_ti2v_720p_121f"
    wl = bench.WORKLOADS[name]
    spec = bench.make_spec(wl)
    dev = torch.device("cuda:0")
    H = wl["H"]
    q, k, v = bench.gen_inputs(wl, H, 0, dev, "iid")
    S = q.shape[2]
    lay = orc.layout_wan(S, wl.get("ffb", 0))
    for f8 in (True, "pv", False):
        call = _core.StagedCall(q, k, v, spec, wl["top_k"], 0.0, None, qkv_fp8=f8, reuse_buffers=False)
        call.select(); call.attend()
        torch.cuda.synchronize()
        print(f"== {name} qkv_fp8={f8}: device O vs the un-quantised dense-masked reference, every query block")
        for h in (0, H // 2, H - 1):
            e = per_block_errors(call, spec, h)
            order = np.argsort(-e[:, 0])[:4]
            print(f"  head {h:2d}: max over blocks {e[:, 0].max():.4f} (block {int(e[:, 0].argmax())}), without the last block "
                  f"{e[:-1, 0].max():.4f}; mean |d| {e[:, 1].mean():.5f}; worst blocks "
                  + ", ".join(f"{int(b)}: {e[b, 0]:.3f} (mean largest P of a row {e[b, 2]:.3f}, R {e[b, 3]:.3f})" for b in order))
        if f8 is not True:
            continue
        # ---- the oracle on head 0: format vs kernel ----
        h = 0
        qh, kh, vh = (x[0, h].float().cpu().numpy() for x in (call.q, call.k, call.v))
        ops = orc.fp8_operands(qh[None, None], kh[None, None], vh[None, None], lay)
        BH = H
        ex_dev = _core.fp8_exps(call.fp8["scales"], BH, spec.NB_total)[h].cpu().numpy().astype(np.uint32)
        km_dev = _core.fp8_kmean(call.fp8["scales"], BH, spec.NB_total, 128)[h].cpu().numpy()
        SP = spec.NB_total * 128
        q8d = call.fp8["q8"].view(BH, SP, 128)[h].cpu().numpy()
        k8d = call.fp8["k8"].view(BH, SP, 128)[h].cpu().numpy()
        v8d = call.fp8["v8t"].view(BH, SP // 64, 128, 64)[h].cpu().numpy()
        print("  operand images of head 0 vs oracle.fp8_operands: exps", np.array_equal(ex_dev, ops["exps"][0]),
              "kmean", np.array_equal(km_dev, ops["kmean"][0]), "q8", np.array_equal(q8d, ops["q8"][0]),
              "k8", np.array_equal(k8d, ops["k8"][0]), "v8t", np.array_equal(v8d, ops["v8t"][0]))
        q8, k8, v8, _ = orc.fp8_dequantized_qkv(qh[None, None], kh[None, None], vh[None, None], lay)
        q8, k8, v8 = q8[0, 0], k8[0, 0], v8[0, 0]
        kept = _core.unpack_bitmask(call.bufs["bitmask"][h:h + 1], spec.NB_total)[0].cpu().numpy()
        R = call.bufs["R"][h].cpu().numpy()
        comp = call.bufs["comp"][h].cpu().numpy()
        NBv = spec.NBv
        blocks = [0, NBv // 2, NBv - 2, NBv - 1]
        km = kept[blocks].astype(np.uint8)
        fin = lambda o: o * R[blocks][:, None, None] + comp[blocks][:, None, :]   # noqa: E731
        ref16 = fin(orc.sparse_attention_head(qh, kh, vh, lay, km, blocks))
        ref8 = fin(orc.sparse_attention_head(q8, k8, v8, lay, km, blocks))
        ref8c = fin(orc.sparse_attention_head_pcode(q8, k8, v8, lay, km, blocks))
        print("  head 0, per block: device vs [un-quantised | e4m3 operands, exact P | e4m3 operands, code-map P];  "
              "oracle(e4m3 operands) vs oracle(un-quantised)")
        for a, i in enumerate(blocks):
            nrow = min(128, S - i * 128)
            got = call.out[0, i * 128:i * 128 + nrow, h].float().cpu().numpy()
            d16, d8, d8c = (np.abs(got - r[a][:nrow]) for r in (ref16, ref8, ref8c))
            fmt = np.abs(ref8[a][:nrow] - ref16[a][:nrow])
            print(f"    block {i:3d} ({nrow:3d} rows, {int(km[a].sum())} kept): {d16.max():.4f}/{d16.mean():.5f} | {d8.max():.4f}/{d8.mean():.5f} | "
                  f"{d8c.max():.4f}/{d8c.mean():.5f} ;  format alone {fmt.max():.4f}/{fmt.mean():.5f}")


if __name__ == "__main__":
    main()
